// bwamem_hip.h — the reference's OUTER boundary over libbwams.so: mem_process_seqs() with the reference's own signature
// (/root/reference/src/bwamem.h:393-395, defined src/bwamem.cpp:1850-1903), for the maintainer who swaps the body of that one
// function.  Inside the reference tree compile with -DBWAMS_HAVE_REFERENCE_HEADERS: the structs then ARE the reference's
// (bwamem.h, bwa.h) and the static_asserts below check that this file's reading of them still holds.  Outside it (this
// repository: ext/safestringlib is not vendored, so bwamem.h does not compile) the layout mirrors below stand in, with the
// sizes and offsets SURVEY.md §8(b) measured on the real headers.
#pragma once
#include <cstddef>
#include <cstdint>
#include "bwams.h"

#ifdef BWAMS_HAVE_REFERENCE_HEADERS
#include "bwamem.h"
#else
// ---- mem_opt_t, src/bwamem.h:89-124 (built without AFF: no start_core) ----
typedef struct mem_opt_t {
    int a, b;
    int o_del, e_del;
    int o_ins, e_ins;
    int pen_unpaired;
    int pen_clip5, pen_clip3;
    int w;
    int zdrop;
    uint64_t max_mem_intv;
    int T;
    int flag;
    int min_seed_len;
    int min_chain_weight;
    int max_chain_extend;
    float split_factor;
    int split_width;
    int max_occ;
    int max_chain_gap;
    int n_threads;
    int64_t chunk_size;
    float mask_level;
    float drop_ratio;
    float XA_drop_ratio;
    float mask_level_redun;
    float mapQ_coef_len;
    int mapQ_coef_fac;
    int max_ins;
    int max_matesw;
    int max_XA_hits, max_XA_hits_alt;
    int8_t mat[25];
} mem_opt_t;
// ---- bseq1_perfect_t, src/perfect.h:131-138; bseq1_t, src/bwa.h:76-86 (OPT_RW and PERFECT_MATCH on: the `scale` build) ----
typedef union {
    struct { uint32_t flags; uint32_t location; };
    uint64_t exist;
} bseq1_perfect_t;
typedef struct {
    int l_seq, id;
    char *strbuf;
    char *name, *comment, *seq, *qual, *sam;
    bseq1_perfect_t perfect;
} bseq1_t;
// ---- mem_pestat_t, src/bwamem.h:178-182 ----
typedef struct {
    int low, high;
    int failed;
    double avg, std;
} mem_pestat_t;
#define MEM_F_PE 0x2
#define MEM_F_NOPAIRING 0x4
#define MEM_F_ALL 0x8
#define MEM_F_NO_MULTI 0x10
#define MEM_F_NO_RESCUE 0x20
#define MEM_F_REF_HDR 0x100
#define MEM_F_SOFTCLIP 0x200
#define MEM_F_SMARTPE 0x400
#define MEM_F_PRIMARY5 0x800
#define MEM_F_KEEP_SUPP_MAPQ 0x1000
#define BATCH_SIZE 512                     /* src/macro.h:63: reads per kt_for work item, i.e. per seqs[].sam string */
#endif

static_assert(sizeof(mem_opt_t) == 176 && offsetof(mem_opt_t, max_mem_intv) == 48 && offsetof(mem_opt_t, T) == 56 &&
              offsetof(mem_opt_t, split_factor) == 76 && offsetof(mem_opt_t, chunk_size) == 96 && offsetof(mem_opt_t, mask_level) == 104 &&
              offsetof(mem_opt_t, mapQ_coef_len) == 120 && offsetof(mem_opt_t, max_XA_hits_alt) == 140 && offsetof(mem_opt_t, mat) == 144,
              "mem_opt_t: 176 bytes, the offsets measured on src/bwamem.h");
static_assert(sizeof(bseq1_t) == 64 && offsetof(bseq1_t, name) == 16 && offsetof(bseq1_t, seq) == 32 && offsetof(bseq1_t, sam) == 48 &&
              offsetof(bseq1_t, perfect) == 56, "bseq1_t of the scale build (OPT_RW, PERFECT_MATCH)");
static_assert(sizeof(mem_pestat_t) == 32 && offsetof(mem_pestat_t, avg) == 16, "mem_pestat_t");
static_assert(sizeof(mem_pestat_t) == sizeof(bwams_pestat_t) && offsetof(bwams_pestat_t, avg) == 16 && offsetof(bwams_pestat_t, failed) == 8,
              "bwams_pestat_t mirrors mem_pestat_t byte for byte");

// What stands where the reference's worker_t holds its FM-index, ERT tables, perfect table and per-thread scratch: the resident
// index set of every GPU and `depth` chunks in flight (a batch per device each, plus page-locked staging for what crosses PCIe every
// chunk).  The pipeline has one mem_process_seqs in flight (src/fastmap.cpp:475-491) but reads chunk i + 1 and writes chunk i - 1
// meanwhile (its `-i` threads): with depth >= 2 the optional _stage / _collect calls below do the same with the GPUs' copies.
// A maintainer keeps one of these beside (or inside) worker_t.
struct bwams_worker;
// idx: bwams_index_from_host / _open at start-up; emf / ert: NULL when not resident; max_reads / max_bases: what
// process() reads per chunk (src/fastmap.cpp:1273-1279).  rg_id: bwa_rg_id (src/bwa.cpp), "" without -R.
int bwams_worker_create(bwams_index_t *idx, bwams_emf_t *emf, bwams_ert_t *ert, int64_t max_reads, int64_t max_bases, const char *rg_id,
                        bwams_worker **out);
void bwams_worker_destroy(bwams_worker *w);
// The same over n_dev GPUs (idx / emf / ert: one handle per device, each on its own replica; emf / ert NULL when not resident) with
// `depth` chunks in flight (1: everything inside mem_process_seqs; 2-3: with the _stage / _collect calls).  A chunk is cut into n_dev
// contiguous shards on read (paired-end: pair) boundaries, mem_pestat's keys are merged in-process, the text returns in read order:
// byte-identical to one device (host/chunk_multi.cpp; tests/test_host_boundary.py).  Every failure is returned, nothing exits.
int bwams_worker_create_multi(bwams_index_t *const *idx, bwams_emf_t *const *emf, bwams_ert_t *const *ert, int n_dev, int depth,
                              int64_t max_reads, int64_t max_bases, const char *rg_id, bwams_worker **out);
const char *bwams_worker_error(const bwams_worker *w);
// set a host path to run for chunks the device path refuses (BWAMS_ERR_UNSUPPORTED); without one such a chunk ends the run
typedef void (*bwams_host_path_t)(mem_opt_t *, int64_t, int, bseq1_t *, const mem_pestat_t *, void *user);
void bwams_worker_set_host_path(bwams_worker *w, bwams_host_path_t f, void *user);

// The reference's signature, worker_t & replaced by the handle above.  Contract kept: reads
// seqs[i].{l_seq, seq, name, qual, comment}, overwrites seqs[i].seq with the base codes, sets seqs[i].perfect when the EMF is
// resident, and leaves ONE malloc'ed SAM string per 512-read work item in seqs[first_of_item].sam (the others NULL), which the
// writer frees (src/fastmap.cpp:437-461).  Errors end the run with a line on stderr, as the reference's do.
void mem_process_seqs(mem_opt_t *opt, int64_t n_processed, int n, bseq1_t *seqs, const mem_pestat_t *pes0, bwams_worker &w);

// The pipeline's two other steps, for a maintainer who lets the GPUs' copies overlap the way the reference overlaps its file I/O
// (three added lines in process(), src/fastmap.cpp:307-468):
//   end of step 0 (the reader's thread):    mem_process_seqs_stage(opt, n, seqs, w)   records -> page-locked arrays -> the devices; waits
//                                           for a free slot (at most `depth` chunks are in flight)
//   step 1:                                 mem_process_seqs(...)                     finds the staged chunk by its seqs pointer
//   start of step 2 (the writer's thread):  mem_process_seqs_collect(opt, n, seqs, w) the SAM strings into seqs[].sam — only after
//                                           bwams_worker_set_deferred_collect(w, 1); otherwise mem_process_seqs has collected already
// Both return 0 or a BWAMS_ERR_* (bwams_worker_error() says why).  Without these calls mem_process_seqs does all three.
int mem_process_seqs_stage(mem_opt_t *opt, int n, bseq1_t *seqs, bwams_worker &w);
int mem_process_seqs_collect(mem_opt_t *opt, int n, bseq1_t *seqs, bwams_worker &w);
void bwams_worker_set_deferred_collect(bwams_worker *w, int on);
// step 1 with an error code instead of the reference's exit(EXIT_FAILURE)
int bwams_worker_process(mem_opt_t *opt, int64_t n_processed, int n, bseq1_t *seqs, const mem_pestat_t *pes0, bwams_worker &w);

// Step 0 of the pipeline for a caller that reads with bwams_reader (include/bwams.h; host/fastq_io.cpp): bseq_read_orig's records
// (src/bwa.cpp:266-335) from a chunk's text IN PLACE — the strings of seqs[i] point into `text`, which is cut up with NULs; wrapped
// records are joined.  copy_comment = 0 drops the comments as process() does without `mem -C`.  Returns the number of records.
int64_t bwams_bseq_parse(char *text, int64_t n_bytes, int64_t n_reads, bseq1_t *seqs, int copy_comment);

// the option mapping on its own (tests compare it with the library's defaults)
void bwams_map_options(const mem_opt_t *opt, const char *rg_id, bwams_seed_opt_t *so, bwams_mem_opt_t *mo, bwams_sam_opt_t *sam_opt);
