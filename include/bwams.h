/*
 * bwams.h — C-ABI of the MI355X-native BWA-MEM seed-and-extend hot path.
 *
 * Plain pointers and sizes only; no C++ or torch types.  Each entry point names
 * the reference interface it replaces (file:line under /root/reference) — these
 * are exactly the call sites a reference maintainer re-binds.
 *
 * Conventions
 *   - every function returns 0 (BWAMS_OK) or a negative error code; the library
 *     never exits the process and never falls back to a CPU path.  A missing or
 *     unusable GPU is BWAMS_ERR_DEVICE.
 *   - the caller owns all host buffers; the library owns device memory.
 *   - output order is input order (SMEMs sorted by (rid, m, n) as
 *     mem_collect_smem leaves them; extension results written back in place).
 *   - a handle is bound to one GPU and one HIP stream; calls on one handle are
 *     serialised by the caller (the reference has one mem_process_seqs call in
 *     flight per pipeline slot, src/fastmap.cpp:475-491).  Use one handle per
 *     host thread / per GPU.
 */
#ifndef BWAMS_H
#define BWAMS_H

#include <stddef.h>
#include <stdint.h>
#include "bwams_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define BWAMS_OK               0
#define BWAMS_ERR_DEVICE      -1   /* no usable gfx950 device / HIP runtime error */
#define BWAMS_ERR_IO          -2   /* index file missing or malformed */
#define BWAMS_ERR_ARG         -3   /* invalid argument */
#define BWAMS_ERR_CAPACITY    -4   /* an output buffer is too small; *n_* holds the need */
#define BWAMS_ERR_NOMEM       -5
#define BWAMS_ERR_UNSUPPORTED -6

typedef struct bwams_index bwams_index_t;   /* FM-index resident in one GPU's HBM */
typedef struct bwams_batch bwams_batch_t;   /* stream + device work buffers for one chunk of reads */
typedef struct bwams_emf bwams_emf_t;       /* exact-match filter table resident in HBM */

const char *bwams_strerror(int code);
/* Text of the last HIP / IO error on this thread. */
const char *bwams_last_error(void);
/* Debugging aids and A-B switches (environment; all result-neutral; read ONCE at the first call into the library, and again by
 * bwams_debug_reload(), which a test calls after changing one):
 *   BWAMS_VERBOSE=1 stage times / counters on stderr    BWAMS_POISON=1 fresh device allocations filled with 0xAB
 *   BWAMS_BWD_MIN_LIST / _COLS / _LATE_LIST, BWAMS_BWD_DRY_MIN_LIST / _DRY_COLS / _DRY_LATE_LIST  when a backward phase of the SMEM search
 *     leaves its lane (entries at the forward end; columns run and entries alive; the same once the read queue is dry; MIN_LIST=0: never)
 *   BWAMS_BWD_FUSED=0, BWAMS_BWD_CAP_MUL, BWAMS_SEED_R3_BESIDE=0, BWAMS_DEBUG   SMEM search launch variants / ablations
 *   BWAMS_EXT_MAX_ROUNDS, BWAMS_EXT_ALL_ROUNDS, BWAMS_EXT_INPLACE=0, BWAMS_BSW_PK=0, BWAMS_CHAIN_BATCH=0   extension rounds and kernel variants
 *   BWAMS_DEDUP_SEQ=1, BWAMS_PAIR_DROP_PLAN, BWAMS_TRACE_PAIR   fallback paths forced by tests; a line per launch of the paired-end tail
 *   BWAMS_ERT_GRID, BWAMS_ERT_FAT=0, BWAMS_ERT_TICKET=0   ERT walk launch shape        BWAMS_HOST_THREADS   host threads of mem_process_seqs' staging (6) */
int bwams_debug_reload(void);
int bwams_device_count(int *n);

/* ---------------------------------------------------------------- index ---- */

/* Host-side description of an FM-index, as FMI_search holds it after
 * load_index (src/FMI_search.cpp:1251-1370): count[] already has the +1. */
typedef struct bwams_fmi_desc {
    int64_t ref_seq_len;               /* 2*l_pac + 1 */
    int64_t count[5];
    const bwams_cp_occ_t *cp_occ;      /* (ref_seq_len >> 6) + 1 blocks */
    const int8_t  *sa_ms_byte;         /* (ref_seq_len >> 3) + 1 */
    const uint32_t *sa_ls_word;        /* (ref_seq_len >> 3) + 1 */
    int64_t sentinel_index;
    const uint8_t *ref_0123;           /* 2*l_pac bytes or NULL (load_ref_string, src/fastmap.cpp:813) */
} bwams_fmi_desc_t;

/* Replaces FMI_search::load_index (src/FMI_search.cpp:1251) + load_ref_string
 * (src/fastmap.cpp:813): reads <prefix>.bwt.2bit.64 (and <prefix>.0123 when
 * present) and uploads them to GPU `device`. */
int bwams_index_open(const char *prefix, int device, bwams_index_t **out);

/* Same, from arrays already in host memory (e.g. the reference's own loaded
 * FMI_search members or its /dev/shm segments, src/bwa_shm.cpp). */
int bwams_index_from_host(const bwams_fmi_desc_t *desc, int device, bwams_index_t **out);

/* Same, adopting arrays that already live in this GPU's memory (not copied, not
 * freed by close).  Lets an on-device index builder hand over without PCIe. */
int bwams_index_from_device(const bwams_fmi_desc_t *desc_with_device_pointers, int device,
                            bwams_index_t **out);

/* FMA direct-lookup tables (all_smem_t / last_smem_t, src/FMI_search.h:101-135; used at
 * src/FMI_search.cpp:1414-1463 and :1705-1750).  build = `bwa-mem2.scale smem-table` on the GPU
 * (one lane per table entry); set = upload tables read from <prefix>.all_smem.11 / .last_smem.13
 * (NULL, NULL detaches them); fetch = download (to write those files).  The reference's depths
 * are all_bp = 11 and last_bp = 13; smaller depths are accepted for testing.  bwams_index_open
 * loads the two files when they exist next to the index. */
int bwams_index_build_fma(bwams_index_t *idx, int all_bp, int last_bp);
int bwams_index_set_fma(bwams_index_t *idx, const void *all_smem, int all_bp, const void *last_smem, int last_bp);
int bwams_index_fetch_fma(bwams_index_t *idx, void *all_smem, void *last_smem);

/* FM-index construction on the GPU.  Replaces `bwa-mem2.scale index`'s FMI_search::build_index +
 * build_fm_index (src/FMI_search.cpp:774-849, :611-771; driver src/main.cpp -> bwa_index, src/bwtindex.cpp):
 * text = fw || revcomp(fw), suffix array (prefix doubling on the device instead of host SA-IS), BWT,
 * CP_OCC, SA samples.  The arrays equal the reference's byte for byte (they are functions of the suffix
 * array).  fw: l_pac base codes 0..3, one per byte (N already replaced, as bns_fasta2bntseq does), in host
 * memory or — fw_on_device != 0 — in this GPU's memory.  Texts up to 2^36 rows (GRCh38: 6.4 G rows, ~150 GB
 * of HBM while building).  chunk_rows bounds the rows sorted at once (0 = 2^30).  stats may be NULL. */
typedef struct bwams_build_stats {
    int64_t rows;                      /* 2*l_pac + 1 */
    int32_t chunks, rounds;            /* key-space chunks of the first pass; doubling rounds after it */
    int64_t unresolved_after_first;    /* rows still tied after the 29-base pass */
    float ms_first_pass, ms_outputs;   /* device time of the first pass / of BWT + CP_OCC + SA samples */
} bwams_build_stats_t;
int bwams_index_build(const uint8_t *fw, int64_t l_pac, int fw_on_device, int device, int keep_ref,
                      int64_t chunk_rows, bwams_build_stats_t *stats, bwams_index_t **out);
/* The resident arrays back to the host (any pointer may be NULL), e.g. to write the reference's files
 * or to hand them to a CPU reader.  desc receives the scalars (count[] with the loader's +1). */
int bwams_index_fetch(bwams_index_t *idx, bwams_cp_occ_t *cp_occ, int8_t *sa_ms_byte, uint32_t *sa_ls_word,
                      uint8_t *ref_0123, bwams_fmi_desc_t *desc);
/* Writes <prefix>.bwt.2bit.64 (and <prefix>.0123 when the index holds the text) in the reference's format
 * (src/FMI_search.cpp:629-763, :796-829), streaming from HBM. */
int bwams_index_save(bwams_index_t *idx, const char *prefix);

int bwams_index_close(bwams_index_t *idx);
int64_t bwams_index_bytes(const bwams_index_t *idx);

/* ---------------------------------------------------------------- batch ---- */

/* Work buffers sized for up to max_reads reads / max_bases bases per call.
 * max_smem / max_sa are the initial sizes of the SMEM and SA-coordinate buffers (0 = library
 * default: 24 SMEMs and 64 coordinates per read on average).  They are not limits: a chunk that
 * needs more makes the stage run a second time on grown buffers (the kernels keep counting when
 * a buffer is full).  The host buffers of bwams_seed_fmi / bwams_seed_fetch are the caller's and
 * still yield BWAMS_ERR_CAPACITY when too small. */
int bwams_batch_create(bwams_index_t *idx, int64_t max_reads, int64_t max_bases,
                       int64_t max_smem, int64_t max_sa, bwams_batch_t **out);
int bwams_batch_destroy(bwams_batch_t *b);

/* -------------------------------------------------------------- seeding ---- */

/* One-call seeding on host buffers.  Replaces, for a whole chunk,
 *   mem_collect_smem            (src/bwamem.cpp:648-786; called at :1321) and
 *   get_sa_entries_prefetch     (src/FMI_search.cpp:2261; called at src/bwamem.cpp:861).
 *
 *   enc_qdb   concatenated reads, one base code per byte (0..3, >=4 = N)
 *   cum_len   nseq+1 offsets into enc_qdb (query_cum_len_ar, widened to 64 bit)
 *   skip      optional nseq flags; non-zero = read resolved by the exact-match
 *             filter, not seeded (seq_[l].perfect.exist, src/bwamem.cpp:674-689)
 *   smem_out  SMEMs of all reads ordered by (rid, m, n); smem_cap slots
 *   sa_coord  reference coordinates of each SMEM's sampled occurrences, in SMEM
 *             order; SMEM i owns sa_coord[sa_off[i] .. sa_off[i+1])
 *   sa_off    n_smem+1 entries (so at least smem_cap+1 slots)
 * sa_coord/sa_off may be NULL to skip the SA step. */
int bwams_seed_fmi(bwams_batch_t *b,
                   const uint8_t *enc_qdb, const int64_t *cum_len, const uint8_t *skip,
                   int64_t nseq, const bwams_seed_opt_t *opt,
                   bwams_smem_t *smem_out, int64_t smem_cap, int64_t *n_smem,
                   int64_t *sa_coord, int64_t sa_cap, int64_t *sa_off, int64_t *n_sa);

/* The same in three steps, so that reads stay resident in HBM across calls and
 * uploads/downloads can overlap other work: upload -> run (asynchronous on the
 * batch's stream) -> fetch (synchronises). */
/* (enc_qdb may be a host pointer or a pointer into this GPU's memory; cum_len and skip are host arrays.) */
int bwams_seed_upload(bwams_batch_t *b, const uint8_t *enc_qdb, const int64_t *cum_len,
                      const uint8_t *skip, int64_t nseq);
int bwams_seed_run(bwams_batch_t *b, const bwams_seed_opt_t *opt, int with_sa);
int bwams_seed_counts(bwams_batch_t *b, int64_t *n_smem, int64_t *n_sa);   /* synchronises */
int bwams_seed_fetch(bwams_batch_t *b, bwams_smem_t *smem_out, int64_t smem_cap,
                     int64_t *sa_coord, int64_t sa_cap, int64_t *sa_off);

/* ------------------------------------------------------------ extension ---- */

/* Banded Smith-Waterman seed extension over n tasks.  Replaces the six call
 * sites BandedPairWiseSW::{scalarBandedSWAWrapper,getScores16,getScores8}
 * (src/bwamem.cpp:3229,3297,3366,3445,3510,3581): fills score, tle, gtle, qle,
 * gscore, max_off of every pair in place with the values of scalarBandedSWA
 * (src/bandedSWA.cpp:116-237).  ref/qer are the flat seqBufRef/seqBufQer byte
 * buffers of ref_bytes/qer_bytes bytes. */
int bwams_bsw_extend(bwams_batch_t *b, bwams_seqpair_t *pairs, int64_t n,
                     const uint8_t *ref, int64_t ref_bytes,
                     const uint8_t *qer, int64_t qer_bytes,
                     int32_t w, const bwams_sw_opt_t *opt);

/* Resident form: upload once, run (async), fetch. */
int bwams_bsw_upload(bwams_batch_t *b, const bwams_seqpair_t *pairs, int64_t n,
                     const uint8_t *ref, int64_t ref_bytes, const uint8_t *qer, int64_t qer_bytes);
int bwams_bsw_run(bwams_batch_t *b, int32_t w, const bwams_sw_opt_t *opt);
int bwams_bsw_fetch(bwams_batch_t *b, bwams_seqpair_t *pairs, int64_t n);

/* ------------------------------------------------------------------- EMF ---- */

/* Exact-match filter.  open/from_host replace load_perfect_table (src/perfect_map.cpp:344):
 * <prefix>.perfect.<L> = 64-byte header, u32 loc_table[], seed_entry_t seed_table[]
 * (src/perfect.h:188-213, :772-822).  The index must hold its .0123 reference. */
int bwams_emf_open(bwams_index_t *idx, const char *path, bwams_emf_t **out);
int bwams_emf_from_host(bwams_index_t *idx, int32_t seed_len, uint32_t seq_len, const uint32_t *loc_table,
                        uint32_t num_loc_entry, const bwams_seed_entry_t *seed_table, uint32_t num_seed_entry,
                        bwams_emf_t **out);
/* adopt a table that already lives in this GPU's memory (not copied, not freed) */
int bwams_emf_from_device(bwams_index_t *idx, int32_t seed_len, uint32_t seq_len, const uint32_t *loc_table_dev,
                          uint32_t num_loc_entry, const bwams_seed_entry_t *seed_table_dev, uint32_t num_seed_entry,
                          bwams_emf_t **out);
/* Builds the table on the GPU from the resident forward reference (replaces `bwa-mem2.scale perfect-index`,
 * src/perfect_index.cpp): every L-mer of the forward strand, canonical orientation, one bucket per hash value with its
 * L-mers in ascending order (root in the bucket's slot, the others in free slots), further locations of repeated L-mers
 * in loc_table.  num_seed_entry = slack x l_pac (the reference uses 1.1).  The table is the reference's format and is
 * probed identically; the placement of collision nodes differs from the reference builder's (which is order dependent). */
int bwams_emf_build(bwams_index_t *idx, int32_t seed_len, double slack, bwams_emf_t **out);
int bwams_emf_info(const bwams_emf_t *emf, int32_t *seed_len, uint32_t *num_seed_entry, uint32_t *num_loc_entry, int64_t *n_used,
                   int64_t *n_key, int64_t *build_ms);
/* the two arrays back to the host (either may be NULL) / the `<prefix>.perfect.<L>` file (src/perfect.h:188-213) */
int bwams_emf_table_fetch(bwams_emf_t *emf, uint32_t *loc_table, bwams_seed_entry_t *seed_table);
int bwams_emf_save(bwams_emf_t *emf, const char *path);
int bwams_emf_close(bwams_emf_t *emf);

/* Replaces the kernel-0 loop of mem_kernel1_core (src/bwamem.cpp:1245-1272): for every read
 * out[i] = seqs[i].perfect and code[i] = the return value of find_perfect_match_entry
 * (src/perfect_map.cpp:638-659; 0 no table / read shorter than L, 1 read has N, 2 not matched,
 * 3 forward match, 4 reverse-complement match, 5 seed matched but not the whole read).
 * `code[i] == 3 || code[i] == 4` is the `skip` flag bwams_seed_fmi takes. */
int bwams_emf_probe(bwams_batch_t *b, bwams_emf_t *emf, const uint8_t *enc_qdb, const int64_t *cum_len,
                    int64_t nseq, bwams_perfect_t *out, uint8_t *code);

/* Resident form: probe the reads uploaded by bwams_seed_upload and set the skip flags on the device,
 * so that the next bwams_seed_run leaves the matched reads out; fetch returns the probe results. */
int bwams_emf_run(bwams_batch_t *b, bwams_emf_t *emf);
int bwams_emf_fetch(bwams_batch_t *b, bwams_perfect_t *out, uint8_t *code);

/* ------------------------------------------------- alignments for SAM (f3) ---- */

/* CIGAR, NM, MD, position and mapping quality of every final region of the chunk.  Replaces, per region, mem_reg2aln
 * (src/bwamem.cpp:2533-2628; called from mem_reg2sam, :2318, and mem_sam_pe, src/bwamem_pair.cpp) with what it calls:
 * bwa_gen_cigar2 (src/bwa.cpp:380-467) -> ksw_global2 with traceback (src/ksw.cpp:558-668), and mem_approx_mapq_se
 * (src/bwamem.cpp:1983-2008).  source 0: the regions bwams_dedup_run left (single-end), 1: those of bwams_pair_run.
 * fetch returns one bwams_aln_t per region in region order, the CIGAR pool (uint32 opLen << 4 | op) and the MD pool
 * (NUL-terminated strings); the XA string and the SAM text stay on the host. */
int bwams_reg2aln_run(bwams_batch_t *b, const bwams_mem_opt_t *opt, int32_t source, int64_t *n_aln, int64_t *n_cigar_ops,
                      int64_t *md_bytes);
/* The same for the regions of bwams_pair_run (source 1), restricted to what the SAM text will read — as the reference, which calls
 * mem_reg2aln from mem_reg2sam / mem_gen_alt / mem_sam_pe only for the records it prints, the members of their XA strings, the paired
 * regions and the mate records (src/bwamem.cpp:2100-2120, src/bwamem_extra.cpp:150-160, src/bwamem_pair.cpp:753-796).  The set follows
 * from the regions, the pairing result and sopt alone; the other regions (two thirds on the bench chunk) keep an unmapped record
 * (rid = -1).  pes = NULL for single-end chunks.  bwams_sam_run / bwams_sam_run_pe with the same sopt then produce the same text as
 * after bwams_reg2aln_run. */
int bwams_reg2aln_run_sam(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_sam_opt_t *sopt, const bwams_pestat_t *pes,
                          int64_t *n_aln, int64_t *n_needed, int64_t *n_cigar_ops, int64_t *md_bytes);
int bwams_reg2aln_fetch(bwams_batch_t *b, bwams_aln_t *aln, int64_t aln_cap, uint32_t *cigar, int64_t cigar_cap, char *md,
                        int64_t md_cap);

/* ------------------------------------------------------ SAM text, single-end ---- *
 * The text worker_sam's single-end branch leaves in seqs[i].sam (src/bwamem.cpp:1836-1844): per read, mem_reg2sam
 * (src/bwamem.cpp:2091-2150: which regions become records, supplementary flag 0x800 / 0x10000, the mapq cap of later records),
 * mem_gen_alt (src/bwamem_extra.cpp:123-187: the XA strings) and mem_aln2sam with m = NULL (src/bwamem.cpp:2380-2531: the
 * eleven fields, NM / MD / AS / XS / RG / SA / pa / XA tags, the comment), with mem_approx_mapq_se (:1983-2008) evaluated on the
 * device.  Call order for a chunk: ... bwams_dedup_run, bwams_pair_run(BWAMS_PAIR_SINGLE_END) (= mem_mark_primary_se),
 * bwams_reg2aln_run(source 1), bwams_sam_upload (names / qualities / comments of the chunk: the bseq1_t fields the hot path
 * never needed), bwams_sam_run, bwams_sam_fetch.  MEM_F_PRIMARY5 (mem_reorder_primary5) acts in bwams_pair_run_sam, MEM_F_REF_HDR (XR:Z:)
 * needs bwams_index_set_contig_annos first; both are built.
 * Paired-end chunks: bwams_sam_run_pe; chunks with reads the EMF resolved: bwams_sam_run_emf (both below). */
/* names of the index's sequences (bntann1_t.name): NUL-terminated, back to back; name_off[n_seqs + 1], name_off[i] = start of
 * name i.  Call after bwams_index_set_contigs (or on a one-sequence index). */
int bwams_index_set_contig_names(bwams_index_t *ix, const char *names, const int32_t *name_off);
/* bntann1_t.anno of the index's sequences, laid out like the names (an empty string = no annotation): what MEM_F_REF_HDR (`mem -V`)
 * prints as XR:Z: (src/bwamem.cpp:2522-2529, :2218-2225).  A text run with that flag and no annotations set is refused. */
int bwams_index_set_contig_annos(bwams_index_t *ix, const char *annos, const int32_t *anno_off);
/* names: the reads' names back to back (no terminators), name_off[nseq + 1]; quals: one byte per base laid out like the reads
 * (cum_len), or NULL ('*'); comments (+ comment_off[nseq + 1], an empty comment = none) or NULL. */
int bwams_sam_upload(bwams_batch_t *b, const char *names, const int64_t *name_off, const char *quals, const char *comments,
                     const int64_t *comment_off);
int bwams_sam_run(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_sam_opt_t *sopt, int64_t *sam_bytes);
/* bwams_sam_run for a chunk that went through the exact-match filter: a read bwams_emf_regs_run resolved in THIS chunk gets the
 * records of mem_perfect2sam_cont / mem_aln2sam_perfect (src/bwamem.cpp:2280-2325, :2153-2227: MAPQ 60, <l_seq>M, NM 0, AS = l_seq * a,
 * XS = AS when there is a second location, secondaries only with MEM_F_ALL, ALT locations last) as worker_sam's `perfect.exist`
 * branch (src/bwamem.cpp:1786-1797) writes them; every other read goes through mem_reg2sam as in bwams_sam_run.  (Paired-end chunks
 * need nothing special: worker_sam turns resolved ends into regions — bwams_emf_regs_run — and calls mem_sam_pe.) */
int bwams_sam_run_emf(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_sam_opt_t *sopt, bwams_emf_t *emf, int64_t *sam_bytes);
/* The paired-end text: mem_sam_pe from the call of mem_pair on (src/bwamem_pair.cpp:686-833 = the tail of mem_sam_pe_batch_post,
 * :1070-1190) for every pair of the chunk (reads 2p, 2p + 1), after bwams_pair_run (mate rescue, marks, mem_pair) and
 * bwams_reg2aln_run(source 1): the multi-hit test, q_pe / q_se with the +40 and tandem-repeat caps, the edits of the paired regions
 * (sub, secondary = -2, the secondary_all switch before XA), the ALT hit, flags 0x1 / 0x2 / 0x8 / 0x20 / 0x40 / 0x80, RNEXT / PNEXT /
 * TLEN, MC:Z, an unmapped end at its mate's coordinates; or the no_pairing branch (proper-pair flag from mem_infer_dir and pes, then
 * mem_reg2sam per end with the mate's record).  pes = what bwams_pestat returned for the chunk.  MEM_F_NOPAIRING is not built. */
int bwams_sam_run_pe(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_sam_opt_t *sopt, const bwams_pestat_t pes[4],
                     int64_t *sam_bytes);
/* sam: the text of all reads in read order (cap >= sam_bytes); read_off[nseq + 1]: where a read's lines start; mapq: the
 * device-side mem_approx_mapq_se of every region (region order of bwams_reg2aln_fetch).  Any of the three may be NULL. */
int bwams_sam_fetch(bwams_batch_t *b, char *sam, int64_t cap, int64_t *read_off, int32_t *mapq, int64_t mapq_cap);

/* ------------------------------------------------------------- read input ---- *
 * A buffer of FASTQ text (host or this GPU's memory) becomes the arrays bwams_seed_upload and bwams_sam_upload take: replaces, per
 * record, kseq_read (src/kseq.h:358-400), trim_readno and kseq2bseq1 (src/bwa.cpp:74-153) as bseq_read_orig (src/bwa.cpp:266-335)
 * calls them, and the base encoding of mem_kernel1_core (src/bwamem.cpp:1232, nst_nt4_table).  Built for FASTQ in the four-lines-per-record
 * shape and for FASTA text (first byte '>', sequences over any number of lines, blank lines skipped; no qualities: bwams_fastq_has_qual = 0);
 * multi-line or mixed FASTQ input, a quality string of another length than its sequence, or a '-' among the bases return
 * BWAMS_ERR_UNSUPPORTED (read such a file on the host).  The buffer must hold whole records; gz decompression and the chunking by
 * base count stay with the caller.  Names come without their "/<digit>" suffix; a read's comment is the header line after the
 * first white-space character (empty = none). */
typedef struct bwams_fastq bwams_fastq_t;
int bwams_fastq_decode(int device, const char *text, int64_t n_bytes, bwams_fastq_t **out, int64_t *n_reads, int64_t *n_bases);
int bwams_fastq_info(const bwams_fastq_t *f, int64_t *n_reads, int64_t *n_bases, int64_t *name_bytes, int64_t *comment_bytes, float *ms);
int bwams_fastq_has_qual(const bwams_fastq_t *f);      /* 0: FASTA text (bseq1_t.qual == NULL for every read) */
/* any pointer may be NULL; enc / quals hold n_bases bytes, cum / name_off / comment_off n_reads + 1 entries */
int bwams_fastq_fetch(bwams_fastq_t *f, uint8_t *enc, int64_t *cum, char *names, int64_t *name_off, char *quals, char *comments,
                      int64_t *comment_off);
/* bwams_seed_upload + bwams_sam_upload of the decoded chunk, device to device.  The reference drops the comments unless `mem -C` was
 * given (process(), src/fastmap.cpp:335-342): copy_comment = 0 does the same; bwams_fastq_to_batch keeps them. */
int bwams_fastq_to_batch(bwams_fastq_t *f, bwams_batch_t *b);
int bwams_fastq_to_batch_opt(bwams_fastq_t *f, bwams_batch_t *b, int32_t copy_comment);
int bwams_fastq_close(bwams_fastq_t *f);

/* ----------------------------------------------------------- mate rescue ---- */

/* Local Smith-Waterman of mate rescue over n tasks: out[i] = ksw_align2(len2, qer + idq,
 * len1, ref + idr, 5, mat, o_del, e_del, o_ins, e_ins, xtra = pairs[i].h0, 0)
 * (src/ksw.cpp:347-381).  Replaces the body of mem_sam_pe_batch
 * (src/bwamem_pair.cpp:880-979: kswv::getScores8/16 phase 0, host-side reversal, phase 1)
 * and the scalar call in mem_matesw (src/bwamem_pair.cpp:217).  The sequence buffers are
 * not modified.  Needs oe_ins + oe_del > max(mat) - min(mat) (true for every bwa-mem
 * scoring scheme), queries <= 512 and targets <= 20000 bases (the row-maxima list of a target must fit one CU's LDS). */
int bwams_ksw_align(bwams_batch_t *b, const bwams_seqpair_t *pairs, int64_t n,
                    const uint8_t *ref, int64_t ref_bytes, const uint8_t *qer, int64_t qer_bytes,
                    const bwams_sw_opt_t *opt, bwams_kswr_t *out);

/* Test hook (not part of the drop-in surface): the region sorts of mem_sort_dedup_patch as the wave tier of the
 * de-duplication kernel runs them, on caller-given keys of n <= 1024 records; order_out[i] = index of the i-th record
 * after the sort.  which: 0 = mem_ars2 (key k = re), 1 = mem_ars (s = score descending, k = rb, q = qb).  mode: 0 = as
 * the kernels choose (rank sort, operation-exact introsort when keys tie), 1 = the wave-parallel operation-exact
 * introsort always, 2 = ksort.h's sequential introsort on one lane (over a copy in global memory); 3 / 4 = modes 1 / 2 with a depth
 * budget of 2, so that the comb-sort fallback of the depth limit sorts nearly everything (the two must agree; the order is not ksort's). */
int bwams_debug_sort(bwams_index_t *idx, const int64_t *k, const int32_t *s, const int32_t *q, int32_t n, int32_t which,
                     int32_t mode, int32_t *order_out);

/* ------------------------------------------------------------- counters ---- */

/* Event counts of the last seed run on this batch (the same events the oracle
 * counts) and per-kernel device times from HIP events recorded
 * on the batch's stream (ms_smem_r1/r2/r3 bracket the search kernel of that round
 * alone).  Synchronises. */
typedef struct bwams_stats {
    int64_t n_ext;            /* backwardExt evaluations */
    int64_t n_ext_blocks;     /* CP_OCC blocks they touch: 1 when k and k+s share a block, else 2 */
    int64_t n_sa_lookups;
    int64_t n_lf_steps;
    int64_t n_smem[3];        /* SMEMs from round 1, 2, 3 */
    int64_t bsw_cells;        /* DP cells evaluated by the last bsw run */
    int64_t n_ext_round[3];   /* n_ext split by round */
    int64_t n_blk_round[3];   /* n_ext_blocks split by round */
    float   ms_smem_r1, ms_smem_r2, ms_smem_r3, ms_sort, ms_sal, ms_seed_total;
    float   ms_bsw;
    float   ms_ksw;
    float   ms_tasks;
    float   ms_emf;
    int64_t emf_nodes;        /* EMF probe: seed entries visited */
    int64_t emf_cmp_bytes;    /* EMF probe: reference bytes compared */
    /* chaining / chain-to-alignment (last bwams_chain_run / bwams_extend_run) */
    int64_t n_chains, n_chain_seeds;
    int64_t n_left, n_right;              /* extension tasks built */
    int64_t n_retry_left, n_retry_right;  /* tasks re-run at twice the band width */
    float   ms_chain, ms_ext_plan, ms_ext_left, ms_ext_right, ms_ext_purge, ms_ext_total;   /* left/right/purge: first round */
    int64_t n_ext_rounds;                 /* extension rounds of the last bwams_extend_run */
    int64_t n_final_regs;                 /* regions left by the last bwams_dedup_run */
    float   ms_dedup;
    float   ms_pair;                      /* last bwams_pair_run, all of it */
    int64_t n_pair_tasks;                 /* rescue alignments (ksw_align2 calls) of the last bwams_pair_run */
    int64_t n_pair_redone;                /* reads whose rescue was redone with every orientation planned */
    int64_t n_pair_regs;                  /* regions after rescue */
    int64_t n_chain_redo;                 /* reads of the last chaining run in which a chain position repeated: chained again
                                           * with the exact B-tree instead of the ordered array */
    /* last bwams_seed_run_ert: events of the walk kernel (ms_smem_r1 = that kernel, ms_smem_r2 = the three rounds over the
     * profiles, ms_smem_r3 = locating the seeds' hits, ms_sal = locating + listing the hits) */
    int64_t ert_kmer_lookups;             /* 8-byte k-mer table entries read */
    int64_t ert_node_reads;               /* tree records decoded (x-mer entry, node head, leaf record / pointer) */
    int64_t ert_ref_bytes;                /* .0123 bytes compared by leaf expansion */
} bwams_stats_t;
int bwams_batch_stats(bwams_batch_t *b, bwams_stats_t *out);

/* ------------------------------------------------------------------------- *
 * ERT seeding: replaces the per-read block of mem_kernel1_core_ert (src/bwamem.cpp:1122-1193: get_seeds /
 * get_seeds_prefix, reseed / reseed_prefix, last, ks_introsort) and the hit sampling of mem_chain_new
 * (src/bwamem.cpp:993-1004), for a whole chunk.
 * ------------------------------------------------------------------------- */
typedef struct bwams_ert bwams_ert_t;

/* The two files `bwa-mem2 index -a ert` writes (src/ertindex.cpp:773-943), as host arrays: kmer_table = 4^kmer_size
 * 8-byte entries, mlt_table = the radix trees.  kmer_size / xmer_size / read_len are the build's kmerSize (15),
 * xmerSize (4) and READ_LEN (src/macro.h:204-206, :66); other values exist for test-sized indexes.  The index handle
 * must hold the .0123 reference (leaf expansion reads it).  The tables are copied into HBM. */
int bwams_ert_from_host(bwams_index_t *idx, const uint64_t *kmer_table, int32_t kmer_size, int32_t xmer_size,
                        int32_t read_len, const uint8_t *mlt_table, int64_t mlt_bytes, bwams_ert_t **out);
/* The same from <prefix>.kmer_table and <prefix>.mlt_table (kmerSize 15, xmerSize 4), streamed into HBM. */
int bwams_ert_open(bwams_index_t *idx, const char *prefix, int32_t read_len, bwams_ert_t **out);
/* Builds the two tables on the GPU from the resident FM-index (replaces buildKmerTrees, src/ertindex.cpp:773-943):
 * the same bytes the reference writes for kmerSize = kmer_size, xmerSize = xmer_size, readLength = read_len and
 * HIT_THRESHOLD = hit_threshold (15, 4, READ_LEN, 256 in src/macro.h).  The index must hold its .0123 reference. */
/* BWAMS_ERR_UNSUPPORTED for a text the format cannot hold (and the reference's writer does not finish on): a string of read_len bases
 * with 65536 occurrences or more (16-bit leaf counts, ertindex.cpp:336-352), a k-mer whose tree reaches 64 MiB (26-bit pointers, :452) */
int bwams_ert_build(bwams_index_t *idx, int32_t kmer_size, int32_t xmer_size, int32_t read_len, int32_t hit_threshold,
                    bwams_ert_t **out);
/* geometry, tree bytes and the build's kernel times (sizes, scan + allocation, bytes; 0 when loaded) */
int bwams_ert_info(const bwams_ert_t *ert, int32_t *kmer_size, int32_t *xmer_size, int32_t *read_len, int64_t *mlt_bytes,
                   float build_ms[3]);
/* copies the tables to the host (either pointer may be NULL) / writes <prefix>.kmer_table and <prefix>.mlt_table */
int bwams_ert_fetch(bwams_ert_t *ert, uint64_t *kmer_table, uint8_t *mlt_table);
int bwams_ert_save(bwams_ert_t *ert, const char *prefix);
int bwams_ert_close(bwams_ert_t *ert);
int64_t bwams_ert_bytes(const bwams_ert_t *ert);
/* The walk's resident entry + tree-head table (64 B per k-mer, derived from the two tables when the handle is made unless BWAMS_ERT_FAT=0:
 * a walk's entry and first records are one line): on = 0 gives its memory back — 64 GiB at k = 15, which a GPU that also holds the EMF
 * wants for its chunks in flight —, on = 1 derives it again.  The results do not depend on it. */
int bwams_ert_set_fat(bwams_ert_t *ert, int32_t on);

/* bwams_seed_run over the ERT instead of the FM-index: same inputs (bwams_seed_upload), same outputs
 * (bwams_seed_counts / bwams_seed_fetch, then bwams_chain_run): the SMEMs of the three seeding rounds in
 * (rid, m, n) order with s = number of hits (k and l are 0: there is no BWT interval), and the sampled hit
 * positions where the FM path puts the suffix-array coordinates.  BWAMS_ERR_UNSUPPORTED when
 * min_seed_len < kmer_size + xmer_size, when split_width + 1 or max_mem_intv exceed 20 (the trees store hit counts
 * below 20 only, src/ertindex.cpp:455-461) or when a read is longer than read_len / 255 bases.
 *
 * Against mem_kernel1_core_ert (bwamem.cpp:1122-1193; restated function by function in oracle/ert_walk_oracle.c): the same MEMs and,
 * per MEM, the same sampled hit coordinates — also for MEMs found by the backward walk and for hits > max_occ, because the
 * reference re-gathers those hits in forward (= suffix-array) order (ertseeding.cpp:644-648) — with ONE exception: a read whose
 * placement at a hit would cross the junction of the forward and the reverse-complement strand of the text.  There get_seq
 * (ertseeding.cpp:455-472) returns nothing and the reference emits matches that are not maximal; this call follows FM-index
 * seeding (the true SMEMs).  Seeds that bridge the junction are dropped by bns_intv2rid in chaining either way. */
int bwams_seed_run_ert(bwams_batch_t *b, bwams_ert_t *ert, const bwams_seed_opt_t *opt, int with_sa);

/* ------------------------------------------------------------------------- *
 * Chaining and chain-to-alignment: replace, for a whole chunk,
 *   mem_chain_seeds + mem_chain_flt (+ the short-read early-out of mem_flt_chained_seeds),
 *     called from mem_kernel1_core, src/bwamem.cpp:1341-1372
 *   mem_chain2aln_across_reads_V2, src/bwamem.cpp:2773-3760, called from mem_kernel2_core
 * ------------------------------------------------------------------------- */

/* The reference sequences of the index (bntseq_t::anns: offset, len, is_alt), needed by
 * bns_intv2rid / bns_fetch_seq_v2.  Without this call the index is one sequence [0, l_pac). */
int bwams_index_set_contigs(bwams_index_t *idx, const bwams_contig_t *contigs, int32_t n_seqs);

/* Chain the seeds the last bwams_seed_run(with_sa = 1) left on the device and filter the chains.
 * Results stay resident; counts are returned.  For reads long enough (5.5 ln L <= 0.05 L, L >= ~1100)
 * mem_flt_chained_seeds' re-scoring of short seeds (mem_seed_sw -> ksw_align2) runs as well; that step
 * needs the index's .0123 reference and, like bwams_ksw_align, oe_ins + oe_del > max(mat) - min(mat). */
int bwams_chain_run(bwams_batch_t *b, const bwams_mem_opt_t *opt, int64_t *n_chains, int64_t *n_seeds);
/* ERT mode: the same stage fed by the reference's ERT walk instead of the FM-index seeding.  Replaces, in
 * mem_kernel1_core_ert (src/bwamem.cpp:1193-1203), ks_introsort(mem_smem_sort_lt) + mem_chain_new (:961-1050) +
 * mem_chain_flt + mem_flt_chained_seeds for the whole chunk; the walk itself (get_seeds / reseed / last,
 * src/ertseeding.cpp) is not built here and stays on the host.  mems: the walk's mem_t records as it leaves them
 * (unsorted), grouped by read (mem_off[nseq + 1]); hits: the reads' hit arrays back to back (hit_off[nseq + 1];
 * mem.hitbeg is relative to its read's slice).  The reads must have been uploaded (bwams_seed_upload).  Results as
 * for bwams_chain_run: bwams_chain_fetch, bwams_extend_run, ... */
int bwams_chain_run_ert(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_ert_mem_t *mems, const int64_t *mem_off,
                        const uint64_t *hits, const int64_t *hit_off, int64_t *n_chains, int64_t *n_seeds);
/* chains grouped by read (chain_off[nseq + 1]) in mem_chain_flt's output order; the seeds of
 * chain c are seeds[c.seed_off .. + c.n) in the order mem_chain_seeds appended them. */
int bwams_chain_fetch(bwams_batch_t *b, bwams_chain_t *chains, int64_t chain_cap, bwams_chain_seed_t *seeds,
                      int64_t seed_cap, int64_t *chain_off);
/* Host-input variant for a caller that keeps chaining on the host: upload chain_ar. */
int bwams_chain_upload(bwams_batch_t *b, const bwams_chain_t *chains, int64_t n_chains, const bwams_chain_seed_t *seeds,
                       int64_t n_seeds, const int64_t *chain_off);

/* Build the extension tasks of the resident chains (phase 1 of mem_chain2aln_across_reads_V2);
 * bwams_extend_run then extends left (band w, retry at 2w), right (h0 = left score), settles the
 * regions and purges covered seeds.  One region per seed, grouped by read (reg_off[nseq + 1]). */
int bwams_extend_build(bwams_batch_t *b, const bwams_mem_opt_t *opt, int64_t *n_left, int64_t *n_right);
int bwams_extend_run(bwams_batch_t *b, const bwams_mem_opt_t *opt, int64_t *n_regs);
int bwams_extend_fetch(bwams_batch_t *b, bwams_alnreg_t *regs, int64_t reg_cap, int64_t *reg_off, int32_t *seed_aln);
/* The tail of mem_kernel2_core (src/bwamem.cpp:1446-1481) on the regions of bwams_extend_run: purged regions
 * dropped, mem_sort_dedup_patch (redundant hits removed, colinear neighbours merged by mem_patch_reg's global
 * alignment, identical hits removed; result ordered by score, rb, qb), the ALT mark.  Needs mask_level_redun in
 * the options.  The extension's regions stay available to bwams_extend_fetch. */
int bwams_dedup_run(bwams_batch_t *b, const bwams_mem_opt_t *opt, int64_t *n_regs);
int bwams_dedup_fetch(bwams_batch_t *b, bwams_alnreg_t *regs, int64_t reg_cap, int64_t *reg_off);
/* mem_perfect2reg with get_perfect_locations and perfect_dedup_patch (src/perfect_map.cpp:659-869), for every read
 * the last bwams_emf_run resolved (code FW_MATCHED / RC_MATCHED): all its exact locations as full-length regions
 * (grouped by read, reg_off[nseq + 1]); first_is_rev[r] = mem_perfect2reg's return value (strand of the first one). */
int bwams_emf_regs_run(bwams_batch_t *b, bwams_emf_t *emf, const bwams_mem_opt_t *opt, int64_t *n_regs);
/* Paired-end chunks: worker_sam gives an end the EMF resolved its regions (mem_perfect2reg, src/bwamem.cpp:1689-1702) before mem_sam_pe.
 * After bwams_dedup_run (and bwams_pestat: mem_pestat ran before, on the regions of worker_aln alone) this appends the regions of
 * bwams_emf_regs_run to the final regions of their reads; bwams_pair_run and what follows then see both. */
int bwams_emf_regs_merge(bwams_batch_t *b, int64_t *n_regs);
int bwams_emf_regs_fetch(bwams_batch_t *b, bwams_alnreg_t *regs, int64_t reg_cap, int64_t *reg_off, uint8_t *first_is_rev);

/* mem_pestat (src/bwamem_pair.cpp:89-156; called at src/bwamem.cpp:1888) over the final regions of
 * bwams_dedup_run: reads 2i and 2i+1 are the two ends of pair i.  pes[4] = orientations FF, FR, RF, RR.
 * The per-pair work (cal_sub, mem_infer_dir) and the sort run on the device; the percentile / mean / std
 * arithmetic over the sorted insert sizes is the reference's sequential double-precision loop, on the host. */
int bwams_pestat(bwams_batch_t *b, const bwams_mem_opt_t *opt, bwams_pestat_t pes[4]);
/* The same in two halves, for a chunk whose pairs are sharded over several batches / GPUs: mem_pestat is a statistic
 * of the WHOLE chunk, the one step of the path where shards exchange data.  bwams_pestat_keys returns one key per
 * qualifying pair of this batch (orientation << 60 | insert size; n_keys <= nseq / 2, BWAMS_ERR_CAPACITY if cap is
 * smaller); the caller concatenates the keys of all shards (an all-gather) and every shard calls
 * bwams_pestat_from_keys (host only, any order of keys) on the union: bit-identical to the unsharded result. */
int bwams_pestat_keys(bwams_batch_t *b, const bwams_mem_opt_t *opt, uint64_t *keys, int64_t cap, int64_t *n_keys);
int bwams_pestat_from_keys(const uint64_t *keys, int64_t n, bwams_pestat_t pes[4]);
/* The paired-end tail of worker_sam up to the pairing decision, for the chunk whose final regions bwams_dedup_run
 * left on the device (reads 2p and 2p + 1 = the ends of pair p):
 *   - mate rescue: mem_sam_pe_batch_pre -> mem_matesw_batch_pre (src/bwamem_pair.cpp:838-870, :1193-1355: anchors
 *     scoring within pen_unpaired of an end's best hit, at most max_matesw; one window per orientation that is not
 *     failed and has no consistent hit yet), the batched ksw_align2 of mem_sam_pe_batch (:880-979), and
 *     mem_sam_pe_batch_post -> mem_matesw_batch_post (:981-1042, :1497-1601: insertion by score and
 *     mem_sort_dedup_patch(opt, 0, 0, 0, ..) after each alignment), or its useErt form (BWAMS_PAIR_USE_ERT);
 *   - mem_mark_primary_se (src/bwamem.cpp:1905-1980) of both ends with ids (id_base + p) << 1 | end;
 *   - mem_pair (src/bwamem_pair.cpp:366-427) when both ends have a primary hit.
 * pes[4] is mem_pestat's result (bwams_pestat) or the caller's (-I); id_base = n_processed >> 1 of the chunk.
 * flags: BWAMS_PAIR_* below.  bwams_pair_fetch returns the regions per read as mem_sam_pe_batch_post holds them
 * before its MAPQ / SAM part (grouped by read, reg_off[nseq + 1]) and one bwams_pair_t per pair.
 * The insert-size term of mem_pair is double arithmetic through log / erfc of the device math library. */
#define BWAMS_PAIR_NO_RESCUE 1   /* MEM_F_NO_RESCUE */
#define BWAMS_PAIR_SINGLE_END 4  /* single-end chunk: just mem_mark_primary_se(opt, n, a, id_base + read) of every read, as mem_reg2sam
                                    does (src/bwamem.cpp:2318-2330); pes may be NULL, no rescue, no pairing, any number of reads */
#define BWAMS_PAIR_USE_ERT   2   /* mem_sam_pe_batch_post's useErt branch (ERT-mode runs): the mate's list is sorted by end
                                  * position, rescue goes through mem_matesw_batch_post_ert (insertion by end, mem_dedup_patch),
                                  * and one mem_sort_dedup_patch or score sort closes each end (src/bwamem_pair.cpp:1017-1041) */
int bwams_pair_run(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_pestat_t pes[4], int64_t id_base, int32_t flags,
                   int64_t *n_regs, int64_t *n_tasks);
/* The same with the flags of mem_opt_t that act between the marking and the SAM text taken from sam_opt->flag:
 *   - MEM_F_PRIMARY5 (`mem -5`): mem_reorder_primary5(sam_opt->T, a) (src/bwamem.cpp:2009-2031) of every read right after
 *     mem_mark_primary_se — in mem_sam_pe before mem_pair (src/bwamem_pair.cpp:1060-1063), in worker_sam's single-end branch
 *     before mem_reg2sam (src/bwamem.cpp:1840);
 *   - MEM_F_NOPAIRING (`mem -P`): mem_pair is not called (src/bwamem_pair.cpp:1066): score 0, z = -1 in every bwams_pair_t;
 *   - MEM_F_NO_RESCUE (`mem -S`): as BWAMS_PAIR_NO_RESCUE.
 * sam_opt == NULL is bwams_pair_run. */
int bwams_pair_run_sam(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_sam_opt_t *sam_opt, const bwams_pestat_t pes[4],
                       int64_t id_base, int32_t flags, int64_t *n_regs, int64_t *n_tasks);
int bwams_pair_fetch(bwams_batch_t *b, bwams_alnreg_t *regs, int64_t reg_cap, int64_t *reg_off, bwams_pair_t *pairs);
/* the task lists as built (side 0 = left, 1 = right), for inspection */
int bwams_extend_tasks_fetch(bwams_batch_t *b, int32_t side, bwams_seqpair_t *pairs, int64_t pair_cap, uint8_t *ref,
                             int64_t ref_cap, uint8_t *qer, int64_t qer_cap, int64_t *n_pairs, int64_t *ref_bytes,
                             int64_t *qer_bytes);

int bwams_batch_sync(bwams_batch_t *b);

/* -------------------------------------------------------- the outer boundary ---- *
 * One chunk, text to text: what kt_pipeline's step 0 parsing (bseq_read_orig, src/bwa.cpp:266-335) and step 1 (mem_process_seqs,
 * src/bwamem.cpp:1850-1980: worker_bwt, worker_aln, mem_pestat, worker_sam) do between the decompressed FASTQ bytes of whole records and
 * seqs[i].sam — the sequence of the stage calls of this header.  fastq: host or device memory, four lines per
 * record, the two ends of a pair interleaved when paired != 0; emf / ert: NULL or the resident tables (ert selects ERT seeding and the
 * useErt form of mate rescue); pes0: NULL = infer the insert-size statistics from the chunk (mem_pestat), as mem_process_seqs does;
 * n_processed: reads processed before this chunk (the hash seeds of mem_mark_primary_se / mem_pair); flags: BWAMS_PAIR_NO_RESCUE
 * (MEM_F_NO_RESCUE) and BWAMS_CHUNK_COPY_COMMENT (`mem -C`: without it the comments of the FASTQ headers are dropped, src/fastmap.cpp:335-342).
 * sam_opt->flag carries the reference's MEM_F_* bits: MEM_F_ALL / NO_MULTI / SOFTCLIP / KEEP_SUPP_MAPQ / REF_HDR act in the text, MEM_F_PRIMARY5 /
 * NOPAIRING / NO_RESCUE before it (bwams_pair_run_sam); MEM_F_PE and MEM_F_SMARTPE are the caller's choice of entry point (paired, _smart).
 * The text stays on the device: bwams_sam_fetch(b, buf, sam_bytes, read_off, NULL, 0) returns it with one offset per read.  The batch must
 * have been created for at least the chunk's reads and bases, the index must carry its sequence names.  Inputs the device path refuses
 * (multi-line FASTQ records, unsupported flags) return BWAMS_ERR_UNSUPPORTED: run that chunk on the host. */
int bwams_process_chunk(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo,
                        const bwams_sam_opt_t *sam_opt, const char *fastq, int64_t n_bytes, int32_t paired, const bwams_pestat_t *pes0,
                        int64_t n_processed, int32_t flags, int64_t *n_reads, int64_t *sam_bytes);
/* The paired-end chunk as two texts (`bwa mem ref R1.fq R2.fq`: bseq_read_orig with a second file, src/bwa.cpp:275-318): record k of
 * fastq1 and record k of fastq2 are the ends of pair k.  Both texts must hold the same number of records (the caller cuts the two files at
 * the same record; the reference stops with a warning when one file runs out). */
int bwams_process_chunk2(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo,
                         const bwams_sam_opt_t *sam_opt, const char *fastq1, int64_t n_bytes1, const char *fastq2, int64_t n_bytes2,
                         const bwams_pestat_t *pes0, int64_t n_processed, int32_t flags, int64_t *n_reads, int64_t *sam_bytes);
/* process()'s MEM_F_SMARTPE branch (`mem -p`, src/fastmap.cpp:378-414): the chunk may mix reads that stand alone with interleaved
 * pairs.  bseq_classify (src/bwa.cpp:346-362: two neighbours carrying one name are a pair, taken greedily from the left) splits it;
 * the single reads go through mem_process_seqs as single-end with ids from n_processed, the pairs as paired-end with ids from
 * n_processed + *n_single and pes0; every read's text returns to its place.  bwams_sam_fetch(b, buf, sam_bytes, read_off, NULL, 0)
 * returns the merged text with *n_reads + 1 offsets (no per-region MAPQ after a merge). */
int bwams_process_chunk_smart(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo,
                              const bwams_sam_opt_t *sam_opt, const char *fastq, int64_t n_bytes, const bwams_pestat_t *pes0,
                              int64_t n_processed, int32_t flags, int64_t *n_reads, int64_t *n_single, int64_t *sam_bytes);
#define BWAMS_CHUNK_COPY_COMMENT 0x100

/* mem_process_seqs (src/bwamem.cpp:1850-1903) for a chunk that arrives the way the reference hands it over: parsed records (bseq1_t:
 * name, comment, seq, qual — src/bwa.h:76-86), here as flat arrays — enc_qdb / cum_len as bwams_seed_upload takes them, names (NUL
 * terminated or not: name_off[i + 1] - name_off[i] bytes each), quals (cum_len's layout, NULL when the reads carry none) and comments as
 * bwams_sam_upload takes them.  The compiled caller with the reference's own signature is bwa-mem-scale_amd/host/mem_process_seqs_hip.cpp.
 * Text out as bwams_process_chunk: bwams_sam_fetch. */
int bwams_process_reads(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo,
                        const bwams_sam_opt_t *sam_opt, const uint8_t *enc_qdb, const int64_t *cum_len, int64_t n_reads, const char *names,
                        const int64_t *name_off, const char *quals, const char *comments, const int64_t *comment_off, int32_t paired,
                        const bwams_pestat_t *pes0, int64_t n_processed, int32_t flags, int64_t *sam_bytes);
/* The same in two halves, for a chunk sharded over several batches (one per GPU; host/chunk_multi.cpp drives them): stage 1 = worker_bwt +
 * worker_aln (up to the regions mem_pestat reads); the caller merges the shards' bwams_pestat_keys with bwams_pestat_from_keys — mem_pestat
 * is a statistic of the WHOLE chunk (bwamem.cpp:1881-1891); stage 2 = worker_sam with the chunk's statistics and the shard's first read id
 * (single-end: n_processed + reads in front of the shard) or pair id (paired-end: (n_processed >> 1) + pairs in front of it). */
int bwams_process_reads_stage1(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo,
                               const uint8_t *enc_qdb, const int64_t *cum_len, int64_t n_reads, const char *names, const int64_t *name_off,
                               const char *quals, const char *comments, const int64_t *comment_off);
int bwams_process_reads_stage2(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_mem_opt_t *mo, const bwams_sam_opt_t *sam_opt,
                               int32_t paired, const bwams_pestat_t *pes, int64_t id_base, int32_t flags, int64_t *sam_bytes);
/* Stage 1 in two halves (bwams_process_reads_stage1 == _upload + _stage1_run): what crosses PCIe, and what computes.  A pipeline puts chunk
 * i + 1 into one batch while another batch of the same device runs chunk i (the reference overlaps reading, computing and writing of
 * consecutive chunks with its `-i` pipeline threads, src/fastmap.cpp:307-468; host/chunk_multi.cpp does the same over batches). */
int bwams_process_reads_upload(bwams_batch_t *b, const uint8_t *enc_qdb, const int64_t *cum_len, int64_t n_reads, const char *names,
                               const int64_t *name_off, const char *quals, const char *comments, const int64_t *comment_off);
int bwams_process_reads_stage1_run(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo);
int bwams_batch_device(const bwams_batch_t *b, int32_t *device);      /* the device of the batch's index */
/* One chunk over n batches — one per GPU, each on a replica of the index — behind one call (host/chunk_multi.cpp): the chunk is cut into n
 * contiguous shards on read (paired-end: pair) boundaries (bwams_shard_bounds: sizes differ by at most one unit, larger shards first), one
 * host thread drives each batch through stage 1, the shards' pestat keys are merged in-process (no collective), stage 2 runs per shard with
 * the chunk's statistics and the shard's first id, and bwams_multi_fetch returns the text in read order with n_reads + 1 offsets.
 * emf / ert: NULL, or one handle per batch (on that batch's device).  Results are byte-identical to one batch processing the whole chunk. */
typedef struct bwams_multi bwams_multi_t;
int bwams_shard_bounds(int64_t n_reads, int32_t n_shards, int32_t paired, int64_t *bounds /* n_shards + 1 */);
int bwams_multi_create(bwams_batch_t *const *batches, bwams_emf_t *const *emf, bwams_ert_t *const *ert, int32_t n, bwams_multi_t **out);
int bwams_multi_process_reads(bwams_multi_t *m, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo, const bwams_sam_opt_t *sam_opt,
                              const uint8_t *enc_qdb, const int64_t *cum_len, int64_t n_reads, const char *names, const int64_t *name_off,
                              const char *quals, const char *comments, const int64_t *comment_off, int32_t paired, const bwams_pestat_t *pes0,
                              int64_t n_processed, int32_t flags, int64_t *sam_bytes);
int bwams_multi_fetch(bwams_multi_t *m, char *sam, int64_t cap, int64_t *read_off);
/* bwams_multi_process_reads in its two halves: _upload cuts the chunk and puts every shard into its batch (the shards' threads, side by
 * side; nothing computes), _compute runs stage 1, the pestat merge and stage 2.  With two bwams_multi over the same devices a caller
 * uploads chunk i + 1 through one while the other computes chunk i (host/mem_process_seqs_hip.cpp: mem_process_seqs_stage / _collect). */
int bwams_multi_upload(bwams_multi_t *m, const uint8_t *enc_qdb, const int64_t *cum_len, int64_t n_reads, const char *names,
                       const int64_t *name_off, const char *quals, const char *comments, const int64_t *comment_off, int32_t paired);
int bwams_multi_compute(bwams_multi_t *m, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo, const bwams_sam_opt_t *sam_opt,
                        const bwams_pestat_t *pes0, int64_t n_processed, int32_t flags, int64_t *sam_bytes);
const char *bwams_multi_error(const bwams_multi_t *m);       /* which shard failed, and why */
int bwams_multi_destroy(bwams_multi_t *m);
/* The file ends of the pipeline (host/fastq_io.cpp).  bwams_reader: a background thread inflates a gz or plain FASTQ / FASTA file (zlib,
 * as the reference's kseq over gzFile) into n_buffers page-locked chunk buffers and cuts chunks where bseq_read_orig cuts them
 * (src/bwa.cpp:266-335): records until the chunk holds chunk_bases bases and, paired, an even number of reads — chunk i holds exactly
 * the reads of the reference's chunk i with the same -K.  _next returns 0 and the chunk (text of whole records; feed it to
 * bwams_process_chunk, or to bwams_bseq_parse + mem_process_seqs), 1 at the end of the file, < 0 on an error (_error says what);
 * the buffer is the caller's until _release.  buffer_bytes 0: 3 bytes per base + 64 MiB.
 * bwams_writer: n_shards output streams ("<path>.<s>.sam"; one stream = `path` itself), each written by its own thread in
 * sequence-number order — the per-GPU output shards behind a sharded job, or the one SAM stream of step 2 (src/fastmap.cpp:437-461). */
typedef struct bwams_reader bwams_reader_t;
typedef struct bwams_writer bwams_writer_t;
int bwams_reader_open(const char *path, int64_t chunk_bases, int32_t paired, int64_t buffer_bytes, int32_t n_buffers, bwams_reader_t **out);
int bwams_reader_next(bwams_reader_t *r, const char **text, int64_t *n_bytes, int64_t *n_reads, int64_t *n_bases);
int bwams_reader_release(bwams_reader_t *r, const char *text);
const char *bwams_reader_error(const bwams_reader_t *r);
int bwams_reader_close(bwams_reader_t *r);
int bwams_writer_open(const char *path, int32_t n_shards, bwams_writer_t **out);
int bwams_writer_put(bwams_writer_t *w, int32_t shard, int64_t seq, const char *text, int64_t n_bytes);
int bwams_writer_close(bwams_writer_t *w);       /* waits until everything handed over in order is on disk */
/* Page-locked host memory (hipHostMalloc) for the buffers that cross PCIe every chunk: reads, names and qualities up, SAM text down. */
int bwams_host_alloc(size_t bytes, void **out);
int bwams_host_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* BWAMS_H */
