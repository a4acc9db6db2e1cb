// api.hip — the C-ABI (include/bwams.h): index residency, batch buffers, and the
// launch sequences of the seed and extend stages on the batch's HIP stream.
//
// There is no CPU fallback anywhere in this file: every entry point either runs
// the HIP kernels on a gfx950 device or returns an error code.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "fmi_kernels.h"
#include "ert_kernels.h"

namespace bwams {

static Knobs g_knobs;
static std::once_flag g_knobs_once;
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; }
void knobs_reload() {
    Knobs k;
    { const char *vb = getenv("BWAMS_VERBOSE"); k.verbose = vb && *vb && *vb != '0'; }
    k.debug = env_int("BWAMS_DEBUG", 0);
    k.poison = env_int("BWAMS_POISON", 0);
    k.bwd_min_list = env_int("BWAMS_BWD_MIN_LIST", k.bwd_min_list); k.bwd_cols = env_int("BWAMS_BWD_COLS", k.bwd_cols);
    k.bwd_late_list = env_int("BWAMS_BWD_LATE_LIST", k.bwd_late_list);
    k.bwd_dry_min_list = env_int("BWAMS_BWD_DRY_MIN_LIST", k.bwd_dry_min_list); k.bwd_dry_cols = env_int("BWAMS_BWD_DRY_COLS", k.bwd_dry_cols);
    k.bwd_dry_late_list = env_int("BWAMS_BWD_DRY_LATE_LIST", k.bwd_dry_late_list);
    k.bwd_fused = env_int("BWAMS_BWD_FUSED", 1); k.bwd_cap_mul = std::max(1, env_int("BWAMS_BWD_CAP_MUL", 1));
    k.r3_beside = env_int("BWAMS_SEED_R3_BESIDE", 1);
    k.ext_max_rounds = env_int("BWAMS_EXT_MAX_ROUNDS", 0); k.ext_all_rounds = getenv("BWAMS_EXT_ALL_ROUNDS") != nullptr;
    k.ext_inplace = env_int("BWAMS_EXT_INPLACE", 1);
    k.dedup_seq = env_int("BWAMS_DEDUP_SEQ", 0) == 1;
    k.pair_drop_plan = getenv("BWAMS_PAIR_DROP_PLAN") != nullptr;
    k.trace_pair = env_int("BWAMS_TRACE_PAIR", 0);
    k.bsw_pk = env_int("BWAMS_BSW_PK", 1);
    k.chain_batch = env_int("BWAMS_CHAIN_BATCH", 1);
    k.cp2 = env_int("BWAMS_CP2", 2);
    k.seed_split = env_int("BWAMS_SEED_SPLIT", 0);
    k.fwd_bpc = std::max(1, env_int("BWAMS_FWD_BPC", 8)); k.bwdl_bpc = std::max(1, env_int("BWAMS_BWDL_BPC", 6));
    k.ert_ticket = env_int("BWAMS_ERT_TICKET", 1); k.ert_grid = env_int("BWAMS_ERT_GRID", -1); k.ert_fat = env_int("BWAMS_ERT_FAT", 1);
    g_knobs = k;
}
const Knobs &knobs() {
    std::call_once(g_knobs_once, knobs_reload);
    return g_knobs;
}

static thread_local std::string g_last_error;
void set_last_error(const std::string &s) { g_last_error = s; }

static int check_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_last_error(std::string("no HIP device: ") + hipGetErrorString(e));
        return BWAMS_ERR_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_last_error("device ordinal out of range");
        return BWAMS_ERR_ARG;
    }
    hipDeviceProp_t p;
    BWAMS_HIP(hipGetDeviceProperties(&p, device));
    if (std::string(p.gcnArchName).rfind("gfx950", 0) != 0) {
        set_last_error(std::string("device is ") + p.gcnArchName + ", this library is built for gfx950 only");
        return BWAMS_ERR_DEVICE;
    }
    return BWAMS_OK;
}

void chain_state_stats(const ChainState *s, bwams_stats_t *out);   // api_chain.hip

int bsw_list_ensure(bwams_batch *b, int64_t n_tasks) {
    if (n_tasks <= b->cap_bsw_list) return BWAMS_OK;
    BWAMS_HIP(hipStreamSynchronize(b->stream));
    if (b->d_bsw_list) (void)hipFree(b->d_bsw_list);
    b->d_bsw_list = nullptr;
    b->cap_bsw_list = n_tasks + n_tasks / 4 + 1024;
    BWAMS_HIP(dev_malloc(&b->d_bsw_list, bsw_list_bytes(b->cap_bsw_list)));
    return BWAMS_OK;
}
int fmi_build_device(bwams_index *ix, const uint8_t *d_fw, int64_t l_pac, int keep_ref, int64_t chunk_rows, int verbose,
                     bwams_build_stats_t *bs);                   // fmi_build.hip

}  // namespace bwams

using namespace bwams;

extern "C" {

const char *bwams_strerror(int code) {
    switch (code) {
        case BWAMS_OK: return "ok";
        case BWAMS_ERR_DEVICE: return "no usable gfx950 device / HIP error";
        case BWAMS_ERR_IO: return "index file missing or malformed";
        case BWAMS_ERR_ARG: return "invalid argument";
        case BWAMS_ERR_CAPACITY: return "output buffer too small";
        case BWAMS_ERR_NOMEM: return "out of memory";
        case BWAMS_ERR_UNSUPPORTED: return "unsupported input";
    }
    return "unknown error";
}

const char *bwams_last_error(void) { return g_last_error.c_str(); }

// the debugging aids and A-B switches are read from the environment once; tests change a variable and call this
int bwams_debug_reload(void) { (void)bwams::knobs(); bwams::knobs_reload(); return BWAMS_OK; }

int bwams_device_count(int *n) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        set_last_error(hipGetErrorString(e));
        return BWAMS_ERR_DEVICE;
    }
    *n = c;
    return BWAMS_OK;
}

/* ------------------------------------------------------------------ index -- */

static int index_finish(bwams_index *ix, const bwams_fmi_desc_t *d) {
    ix->fmi.cp = reinterpret_cast<const uint4 *>(ix->d_cp);
    ix->fmi.cp2 = nullptr;
    ix->fmi.tab_kind = 0;
    ix->fmi.sa_ms = reinterpret_cast<const int8_t *>(ix->d_ms);
    ix->fmi.sa_ls = reinterpret_cast<const uint32_t *>(ix->d_ls);
    ix->fmi.ref = reinterpret_cast<const uint8_t *>(ix->d_ref);
    if (d->ref_seq_len >= ((int64_t)1 << 36)) {
        set_last_error("text longer than 2^36 rows is not supported by the 36-bit interval packing");
        return BWAMS_ERR_UNSUPPORTED;
    }
    for (int i = 0; i < 5; ++i) ix->fmi.count[i] = d->count[i];
    ix->fmi.sentinel = d->sentinel_index;
    ix->fmi.ref_seq_len = d->ref_seq_len;
    return BWAMS_OK;
}

int bwams_index_from_host(const bwams_fmi_desc_t *d, int device, bwams_index_t **out) {
    if (!d || !out || !d->cp_occ || !d->sa_ms_byte || !d->sa_ls_word || d->ref_seq_len <= 0) {
        set_last_error("bwams_index_from_host: null or empty descriptor");
        return BWAMS_ERR_ARG;
    }
    int rc = check_device(device);
    if (rc) return rc;
    BWAMS_HIP(hipSetDevice(device));
    bwams_index *ix = new bwams_index();
    ix->device = device;
    ix->n_blk = (d->ref_seq_len >> 6) + 1;
    ix->n_sa = (d->ref_seq_len >> 3) + 1;
    const size_t b_cp = (size_t)ix->n_blk * 64, b_ms = (size_t)ix->n_sa, b_ls = (size_t)ix->n_sa * 4;
    const size_t b_ref = d->ref_0123 ? (size_t)(d->ref_seq_len - 1) : 0;
    // a failed allocation or copy must not strand the multi-GB buffers already made: close the handle on the way out
    auto up = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = dev_malloc(dst, bytes + 64);          // slack: kernels read whole aligned words
        return e != hipSuccess ? e : hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    hipError_t ue = up(&ix->d_cp, d->cp_occ, b_cp);
    if (ue == hipSuccess) ue = up(&ix->d_ms, d->sa_ms_byte, b_ms);
    if (ue == hipSuccess) ue = up(&ix->d_ls, d->sa_ls_word, b_ls);
    if (ue == hipSuccess && b_ref) ue = up(&ix->d_ref, d->ref_0123, b_ref);
    if (ue != hipSuccess) {
        bwams_index_close(ix);
        BWAMS_HIP(ue);
    }
    ix->bytes = (int64_t)(b_cp + b_ms + b_ls + b_ref);
    int frc = index_finish(ix, d);
    if (frc) { bwams_index_close(ix); return frc; }
    *out = ix;
    return BWAMS_OK;
}

int bwams_index_from_device(const bwams_fmi_desc_t *d, int device, bwams_index_t **out) {
    if (!d || !out || !d->cp_occ || !d->sa_ms_byte || !d->sa_ls_word || d->ref_seq_len <= 0) {
        set_last_error("bwams_index_from_device: null or empty descriptor");
        return BWAMS_ERR_ARG;
    }
    int rc = check_device(device);
    if (rc) return rc;
    bwams_index *ix = new bwams_index();
    ix->device = device;
    ix->owns = false;
    ix->n_blk = (d->ref_seq_len >> 6) + 1;
    ix->n_sa = (d->ref_seq_len >> 3) + 1;
    ix->d_cp = const_cast<bwams_cp_occ_t *>(d->cp_occ);
    ix->d_ms = const_cast<int8_t *>(d->sa_ms_byte);
    ix->d_ls = const_cast<uint32_t *>(d->sa_ls_word);
    ix->d_ref = const_cast<uint8_t *>(d->ref_0123);
    ix->bytes = ix->n_blk * 64 + ix->n_sa * 5 + (d->ref_0123 ? d->ref_seq_len - 1 : 0);
    int frc = index_finish(ix, d);
    if (frc) { bwams_index_close(ix); return frc; }
    *out = ix;
    return BWAMS_OK;
}

int bwams_index_open(const char *prefix, int device, bwams_index_t **out) {
    if (!prefix || !out) return BWAMS_ERR_ARG;
    std::string path = std::string(prefix) + ".bwt.2bit.64";
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) {
        set_last_error("cannot open " + path);
        return BWAMS_ERR_IO;
    }
    struct stat st;
    fstat(fd, &st);
    const size_t fsz = (size_t)st.st_size;
    if (fsz < 56) {
        close(fd);
        set_last_error(path + ": truncated");
        return BWAMS_ERR_IO;
    }
    const uint8_t *m = (const uint8_t *)mmap(nullptr, fsz, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        set_last_error("mmap failed: " + path);
        return BWAMS_ERR_IO;
    }
    bwams_fmi_desc_t d;
    memset(&d, 0, sizeof d);
    memcpy(&d.ref_seq_len, m, 8);
    int64_t cnt[5];
    memcpy(cnt, m + 8, 40);
    for (int i = 0; i < 5; ++i) d.count[i] = cnt[i] + 1;   // as the reference loader does (FMI_search.cpp:880-883)
    const int64_t n_blk = (d.ref_seq_len >> 6) + 1, n_sa = (d.ref_seq_len >> 3) + 1;
    const size_t need = 48 + (size_t)n_blk * 64 + (size_t)n_sa * 5 + 8;
    if (d.ref_seq_len <= 0 || fsz != need) {
        munmap((void *)m, fsz);
        set_last_error(path + ": size does not match its header");
        return BWAMS_ERR_IO;
    }
    size_t o = 48;
    d.cp_occ = reinterpret_cast<const bwams_cp_occ_t *>(m + o);
    o += (size_t)n_blk * 64;
    d.sa_ms_byte = reinterpret_cast<const int8_t *>(m + o);
    o += (size_t)n_sa;
    // sa_ls_word is not 4-byte aligned in the file in general: stage through an aligned copy
    std::vector<uint32_t> ls((size_t)n_sa);
    memcpy(ls.data(), m + o, (size_t)n_sa * 4);
    d.sa_ls_word = ls.data();
    o += (size_t)n_sa * 4;
    memcpy(&d.sentinel_index, m + o, 8);

    // optional .0123
    std::string rpath = std::string(prefix) + ".0123";
    const uint8_t *rm = nullptr;
    size_t rsz = 0;
    int rfd = open(rpath.c_str(), O_RDONLY);
    if (rfd >= 0) {
        struct stat rs;
        fstat(rfd, &rs);
        rsz = (size_t)rs.st_size;
        if (rsz == (size_t)(d.ref_seq_len - 1)) {
            rm = (const uint8_t *)mmap(nullptr, rsz, PROT_READ, MAP_PRIVATE, rfd, 0);
            if (rm == MAP_FAILED) rm = nullptr;
        }
        close(rfd);
    }
    d.ref_0123 = rm;
    int rc = bwams_index_from_host(&d, device, out);
    if (rm) munmap((void *)rm, rsz);
    munmap((void *)m, fsz);
    if (rc) return rc;
    // optional FMA tables written by `bwa-mem2.scale smem-table` (src/FMI_search.cpp:228-277)
    {
        std::string pa = std::string(prefix) + ".all_smem.11", pl = std::string(prefix) + ".last_smem.13";
        int fa = open(pa.c_str(), O_RDONLY), fl = open(pl.c_str(), O_RDONLY);
        struct stat sa, sl;
        if (fa >= 0 && fl >= 0 && fstat(fa, &sa) == 0 && fstat(fl, &sl) == 0 &&
            (size_t)sa.st_size == ((size_t)1 << 22) * 128 && (size_t)sl.st_size == ((size_t)1 << 26) * 16) {
            void *ma = mmap(nullptr, (size_t)sa.st_size, PROT_READ, MAP_PRIVATE, fa, 0);
            void *ml = mmap(nullptr, (size_t)sl.st_size, PROT_READ, MAP_PRIVATE, fl, 0);
            if (ma != MAP_FAILED && ml != MAP_FAILED) rc = bwams_index_set_fma(*out, ma, 11, ml, 13);
            if (ma != MAP_FAILED) munmap(ma, (size_t)sa.st_size);
            if (ml != MAP_FAILED) munmap(ml, (size_t)sl.st_size);
        }
        if (fa >= 0) close(fa);
        if (fl >= 0) close(fl);
    }
    return rc;
}

int bwams_index_build(const uint8_t *fw, int64_t l_pac, int fw_on_device, int device, int keep_ref, int64_t chunk_rows,
                      bwams_build_stats_t *stats, bwams_index_t **out) {
    if (!fw || !out || l_pac <= 0) {
        set_last_error("bwams_index_build: null or empty sequence");
        return BWAMS_ERR_ARG;
    }
    int rc = check_device(device);
    if (rc) return rc;
    BWAMS_HIP(hipSetDevice(device));
    void *staged = nullptr;
    if (!fw_on_device) {
        BWAMS_HIP(dev_malloc(&staged, (size_t)l_pac));
        hipError_t e = hipMemcpy(staged, fw, (size_t)l_pac, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(staged); BWAMS_HIP(e); }
    }
    bwams_index *ix = new bwams_index();
    ix->device = device;
    rc = fmi_build_device(ix, staged ? (const uint8_t *)staged : fw, l_pac, keep_ref, chunk_rows, knobs().verbose != 0, stats);
    if (staged) (void)hipFree(staged);
    if (rc) { bwams_index_close(ix); return rc; }
    *out = ix;
    return BWAMS_OK;
}

int bwams_index_fetch(bwams_index_t *ix, bwams_cp_occ_t *cp_occ, int8_t *sa_ms_byte, uint32_t *sa_ls_word, uint8_t *ref_0123,
                      bwams_fmi_desc_t *d) {
    if (!ix) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(ix->device));
    if (cp_occ) BWAMS_HIP(hipMemcpy(cp_occ, ix->d_cp, (size_t)ix->n_blk * 64, hipMemcpyDeviceToHost));
    if (sa_ms_byte) BWAMS_HIP(hipMemcpy(sa_ms_byte, ix->d_ms, (size_t)ix->n_sa, hipMemcpyDeviceToHost));
    if (sa_ls_word) BWAMS_HIP(hipMemcpy(sa_ls_word, ix->d_ls, (size_t)ix->n_sa * 4, hipMemcpyDeviceToHost));
    if (ref_0123) {
        if (!ix->d_ref) {
            set_last_error("bwams_index_fetch: the index holds no .0123 text");
            return BWAMS_ERR_ARG;
        }
        BWAMS_HIP(hipMemcpy(ref_0123, ix->d_ref, (size_t)(ix->fmi.ref_seq_len - 1), hipMemcpyDeviceToHost));
    }
    if (d) {
        memset(d, 0, sizeof *d);
        d->ref_seq_len = ix->fmi.ref_seq_len;
        for (int i = 0; i < 5; ++i) d->count[i] = ix->fmi.count[i];
        d->sentinel_index = ix->fmi.sentinel;
    }
    return BWAMS_OK;
}

int bwams_index_save(bwams_index_t *ix, const char *prefix) {
    if (!ix || !prefix) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(ix->device));
    const size_t kSlab = (size_t)256 << 20;
    std::vector<uint8_t> slab(kSlab);
    auto stream_out = [&](FILE *f, const void *dev, size_t bytes) -> int {
        for (size_t o = 0; o < bytes; o += kSlab) {
            const size_t n = std::min(kSlab, bytes - o);
            BWAMS_HIP(hipMemcpy(slab.data(), (const uint8_t *)dev + o, n, hipMemcpyDeviceToHost));
            if (fwrite(slab.data(), 1, n, f) != n) return BWAMS_ERR_IO;
        }
        return BWAMS_OK;
    };
    std::string path = std::string(prefix) + ".bwt.2bit.64";
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) {
        set_last_error("cannot create " + path);
        return BWAMS_ERR_IO;
    }
    int64_t hdr[6];
    hdr[0] = ix->fmi.ref_seq_len;
    for (int i = 0; i < 5; ++i) hdr[1 + i] = ix->fmi.count[i] - 1;          // the file holds them without the loader's +1
    int rc = fwrite(hdr, 8, 6, f) == 6 ? BWAMS_OK : BWAMS_ERR_IO;
    if (!rc) rc = stream_out(f, ix->d_cp, (size_t)ix->n_blk * 64);
    if (!rc) rc = stream_out(f, ix->d_ms, (size_t)ix->n_sa);
    if (!rc) rc = stream_out(f, ix->d_ls, (size_t)ix->n_sa * 4);
    const int64_t sent = ix->fmi.sentinel;
    if (!rc && fwrite(&sent, 8, 1, f) != 1) rc = BWAMS_ERR_IO;
    if (fclose(f) != 0 && !rc) rc = BWAMS_ERR_IO;
    if (!rc && ix->d_ref) {
        path = std::string(prefix) + ".0123";
        f = fopen(path.c_str(), "wb");
        if (!f) rc = BWAMS_ERR_IO;
        else {
            rc = stream_out(f, ix->d_ref, (size_t)(ix->fmi.ref_seq_len - 1));
            if (fclose(f) != 0 && !rc) rc = BWAMS_ERR_IO;
        }
    }
    if (rc == BWAMS_ERR_IO) set_last_error("write failed: " + path);
    return rc;
}

int bwams_index_close(bwams_index_t *ix) {
    if (!ix) return BWAMS_OK;
    if (ix->d_cp2) { (void)hipSetDevice(ix->device); (void)hipFree(ix->d_cp2); ix->d_cp2 = nullptr; }
    if (ix->owns) {
        (void)hipSetDevice(ix->device);
        (void)hipFree(ix->d_cp);
        (void)hipFree(ix->d_ms);
        (void)hipFree(ix->d_ls);
        if (ix->d_ref) (void)hipFree(ix->d_ref);
    }
    if (ix->d_contigs) {
        (void)hipSetDevice(ix->device);
        (void)hipFree(ix->d_contigs);
    }
    if (ix->d_ctg_annos) {
        (void)hipFree(ix->d_ctg_annos);
        (void)hipFree(ix->d_ctg_anno_off);
    }
    if (ix->d_ctg_names) {
        (void)hipSetDevice(ix->device);
        (void)hipFree(ix->d_ctg_names);
        (void)hipFree(ix->d_ctg_off);
    }
    if (ix->d_all || ix->d_last) {
        (void)hipSetDevice(ix->device);
        if (ix->d_all) (void)hipFree(ix->d_all);
        if (ix->d_last) (void)hipFree(ix->d_last);
    }
    delete ix;
    return BWAMS_OK;
}

int64_t bwams_index_bytes(const bwams_index_t *ix) { return ix ? ix->bytes : 0; }

/* ------------------------------------------------------------------- FMA ---- */

static int fma_alloc(bwams_index *ix, int all_bp, int last_bp) {
    if (all_bp < 2 || all_bp > 11 || last_bp < 2 || last_bp > 13) {
        set_last_error("FMA depths must be 2..11 (all_smem) and 2..13 (last_smem)");
        return BWAMS_ERR_ARG;
    }
    BWAMS_HIP(hipSetDevice(ix->device));
    if (ix->d_all) (void)hipFree(ix->d_all);
    if (ix->d_last) (void)hipFree(ix->d_last);
    ix->d_all = ix->d_last = nullptr;
    ix->fmi.all_smem = nullptr;
    ix->fmi.last_smem = nullptr;
    BWAMS_HIP(dev_malloc(&ix->d_all, ((size_t)1 << (2 * all_bp)) * 128));
    BWAMS_HIP(dev_malloc(&ix->d_last, ((size_t)1 << (2 * last_bp)) * 16));
    return BWAMS_OK;
}

static void fma_attach(bwams_index *ix, int all_bp, int last_bp) {
    ix->fmi.all_smem = reinterpret_cast<const uint32_t *>(ix->d_all);
    ix->fmi.last_smem = reinterpret_cast<const uint4 *>(ix->d_last);
    ix->fmi.all_bp = all_bp;
    ix->fmi.last_bp = last_bp;
}

int bwams_index_build_fma(bwams_index_t *ix, int all_bp, int last_bp) {
    if (!ix) return BWAMS_ERR_ARG;
    int rc = fma_alloc(ix, all_bp, last_bp);
    if (rc) return rc;
    launch_build_fma(ix->fmi, all_bp, reinterpret_cast<uint32_t *>(ix->d_all), last_bp,
                     reinterpret_cast<uint4 *>(ix->d_last), nullptr);
    BWAMS_HIP(hipGetLastError());
    BWAMS_HIP(hipDeviceSynchronize());
    fma_attach(ix, all_bp, last_bp);
    return BWAMS_OK;
}

int bwams_index_set_fma(bwams_index_t *ix, const void *all_smem, int all_bp, const void *last_smem, int last_bp) {
    if (!ix) return BWAMS_ERR_ARG;
    if (!all_smem || !last_smem) {                         // detach: FM-index only
        ix->fmi.all_smem = nullptr;
        ix->fmi.last_smem = nullptr;
        return BWAMS_OK;
    }
    int rc = fma_alloc(ix, all_bp, last_bp);
    if (rc) return rc;
    BWAMS_HIP(hipMemcpy(ix->d_all, all_smem, ((size_t)1 << (2 * all_bp)) * 128, hipMemcpyHostToDevice));
    BWAMS_HIP(hipMemcpy(ix->d_last, last_smem, ((size_t)1 << (2 * last_bp)) * 16, hipMemcpyHostToDevice));
    fma_attach(ix, all_bp, last_bp);
    return BWAMS_OK;
}

int bwams_index_fetch_fma(bwams_index_t *ix, void *all_smem, void *last_smem) {
    if (!ix || !ix->d_all || !ix->d_last) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(ix->device));
    if (all_smem) BWAMS_HIP(hipMemcpy(all_smem, ix->d_all, ((size_t)1 << (2 * ix->fmi.all_bp)) * 128, hipMemcpyDeviceToHost));
    if (last_smem) BWAMS_HIP(hipMemcpy(last_smem, ix->d_last, ((size_t)1 << (2 * ix->fmi.last_bp)) * 16, hipMemcpyDeviceToHost));
    return BWAMS_OK;
}

/* ------------------------------------------------------------------ batch -- */

// (re)allocate every buffer whose size follows max_smem; the batch grows them when a chunk needs more
static int alloc_smem_buffers(bwams_batch *b, int64_t max_smem) {
    void **ptrs[] = {(void **)&b->d_pool, (void **)&b->d_pool3, (void **)&b->d_sorted, (void **)&b->d_keys, (void **)&b->d_keys2, (void **)&b->d_vals,
                     (void **)&b->d_vals2, (void **)&b->d_work2, (void **)&b->d_sa_off, (void **)&b->d_sa_cnt};
    for (void **p : ptrs)
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    b->max_smem = max_smem;
    // the pool is handed out in per-wave chunks: room for every wave's partly filled last chunk
    // of each of the five emitting launches on top of the max_smem real records
    b->pool_cap = b->max_smem + seed_pool_slack(b->cu_count);
    BWAMS_HIP(dev_malloc(&b->d_pool, (size_t)b->pool_cap * sizeof(bwams_smem_t)));
    // round 3's own pool (it runs from the start of the stage): it may hold most of a chunk's records (a clean unique read has one
    // SMEM and half a dozen round-3 seeds), so it is as large as the main one's record part + one launch's chunk tails
    b->pool3_cap = b->max_smem + seed_pool_slack(b->cu_count) / 5;
    if (knobs().r3_beside == 2) BWAMS_HIP(dev_malloc(&b->d_pool3, (size_t)b->pool3_cap * sizeof(bwams_smem_t)));      // the experiment's buffer: only on demand
    BWAMS_HIP(dev_malloc(&b->d_sorted, (size_t)b->max_smem * sizeof(bwams_smem_t)));
    BWAMS_HIP(dev_malloc(&b->d_keys, (size_t)b->pool_cap * 8));
    BWAMS_HIP(dev_malloc(&b->d_keys2, (size_t)b->pool_cap * 8));
    BWAMS_HIP(dev_malloc(&b->d_vals, (size_t)b->pool_cap * 4));
    BWAMS_HIP(dev_malloc(&b->d_vals2, (size_t)b->pool_cap * 4));
    BWAMS_HIP(dev_malloc(&b->d_work2, (size_t)b->pool_cap * sizeof(Round2Work)));
    BWAMS_HIP(dev_malloc(&b->d_sa_off, (size_t)(b->max_smem + 1) * 8));
    BWAMS_HIP(dev_malloc(&b->d_sa_cnt, (size_t)(b->max_smem + 1) * 8));
    return BWAMS_OK;
}

// everything of bwams_batch_create that can fail half way: the caller destroys the handle (streams, events and the buffers made so
// far) when this returns an error
static int batch_create_fill(bwams_batch *b, bwams_index_t *ix, int64_t max_reads, int64_t max_bases, int64_t max_smem, int64_t max_sa) {
    b->idx = ix;
    b->max_reads = max_reads;
    b->max_bases = max_bases;
    b->max_smem = max_smem > 0 ? max_smem : 24 * max_reads + 1024;
    b->max_sa = max_sa > 0 ? max_sa : 64 * max_reads + 1024;
    hipDeviceProp_t prop;
    BWAMS_HIP(hipGetDeviceProperties(&prop, ix->device));
    b->cu_count = prop.multiProcessorCount;
    BWAMS_HIP(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    BWAMS_HIP(hipStreamCreateWithFlags(&b->seed_aux, hipStreamNonBlocking));
    BWAMS_HIP(hipEventCreateWithFlags(&b->seed_fork, hipEventDisableTiming));
    BWAMS_HIP(hipEventCreateWithFlags(&b->seed_join, hipEventDisableTiming));
    for (auto &e : b->ev) BWAMS_HIP(hipEventCreate(&e));
    for (auto &e : b->ev_emf) BWAMS_HIP(hipEventCreate(&e));

    BWAMS_HIP(dev_malloc(&b->d_enc, (size_t)max_bases + 64));
    BWAMS_HIP(dev_malloc(&b->d_cum, (size_t)(max_reads + 1) * 8));
    BWAMS_HIP(dev_malloc(&b->d_skip, (size_t)max_reads));
    if (int arc = alloc_smem_buffers(b, b->max_smem)) return arc;
    BWAMS_HIP(dev_malloc(&b->d_sa_coord, (size_t)b->max_sa * 8));
    BWAMS_HIP(dev_malloc(&b->d_ctr, sizeof(DevCounters)));
    BWAMS_HIP(dev_malloc(&b->d_ctr3, sizeof(DevCounters)));
    BWAMS_HIP(hipHostMalloc(&b->h_ctr, sizeof(DevCounters)));
    BWAMS_HIP(hipMemset(b->d_ctr, 0, sizeof(DevCounters)));

    // rocPRIM temporary storage for the largest sort / scan this batch can issue
    size_t t1 = 0, t2 = 0;
    (void)rocprim::radix_sort_pairs(nullptr, t1, b->d_keys, b->d_keys2, b->d_vals, b->d_vals2,
                              (size_t)b->max_smem, 0, 64, b->stream);
    (void)rocprim::exclusive_scan(nullptr, t2, b->d_sa_cnt, b->d_sa_off, (int64_t)0, (size_t)b->max_smem + 1,
                            rocprim::plus<int64_t>(), b->stream);
    b->tmp_bytes = std::max(t1, t2);
    BWAMS_HIP(dev_malloc(&b->d_tmp, b->tmp_bytes));
    return BWAMS_OK;
}

int bwams_batch_create(bwams_index_t *ix, int64_t max_reads, int64_t max_bases, int64_t max_smem,
                       int64_t max_sa, bwams_batch_t **out) {
    if (!ix || !out || max_reads <= 0 || max_bases <= 0) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(ix->device));
    bwams_batch *b = new bwams_batch();
    const int rc = batch_create_fill(b, ix, max_reads, max_bases, max_smem, max_sa);
    if (rc) {                                   // (the message of the failing call stays in bwams_last_error)
        b->idx = ix;
        bwams_batch_destroy(b);
        return rc;
    }
    *out = b;
    return BWAMS_OK;
}

int bwams_batch_destroy(bwams_batch_t *b) {
    if (!b) return BWAMS_OK;
    (void)hipSetDevice(b->idx->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    void *ptrs[] = {b->d_enc, b->d_cum, b->d_skip, b->d_pool, b->d_sorted, b->d_keys, b->d_keys2, b->d_vals,
                    b->d_vals2, b->d_work2, b->d_sa_off, b->d_sa_cnt, b->d_sa_coord, b->d_tmp, b->d_ctr, b->d_ctr3, b->d_pool3, b->d_prev, b->d_packed, b->d_emf_out, b->d_emf_code, b->d_ksw_out, b->d_bsw_list, b->d_pairs, b->d_ref, b->d_qer, b->d_ert_prof, b->d_ert_stk, b->d_ert_redo, b->d_bwd_items, b->d_bwd_ent, b->d_f_items, b->d_fl_ent};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (b->h_ctr) (void)hipHostFree(b->h_ctr);
    if (b->chain) chain_state_free(b->chain);
    for (auto &e : b->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : b->ev_emf)
        if (e) (void)hipEventDestroy(e);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    if (b->seed_aux) (void)hipStreamDestroy(b->seed_aux);
    if (b->seed_fork) (void)hipEventDestroy(b->seed_fork);
    if (b->seed_join) (void)hipEventDestroy(b->seed_join);
    delete b;
    return BWAMS_OK;
}

int bwams_batch_sync(bwams_batch_t *b) {
    if (!b) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipStreamSynchronize(b->stream));
    return BWAMS_OK;
}

/* ---------------------------------------------------------------- seeding -- */

int bwams_seed_upload(bwams_batch_t *b, const uint8_t *enc, const int64_t *cum, const uint8_t *skip,
                      int64_t nseq) {
    if (!b || !enc || !cum || nseq < 0) return BWAMS_ERR_ARG;
    if (nseq > b->max_reads) {
        set_last_error("bwams_seed_upload: more reads than the batch was created for");
        return BWAMS_ERR_CAPACITY;
    }
    const int64_t base0 = cum[0];
    const int64_t nb = cum[nseq] - base0;
    if (base0 != 0 || nb > b->max_bases || nb < 0) {
        set_last_error("bwams_seed_upload: cum_len must start at 0 and fit max_bases");
        return nb > b->max_bases ? BWAMS_ERR_CAPACITY : BWAMS_ERR_ARG;
    }
    int mx = 0;
    for (int64_t i = 0; i < nseq; ++i) {
        const int64_t l = cum[i + 1] - cum[i];
        if (l < 0 || l > 0xfffe) {
            set_last_error("bwams_seed_upload: read length must be in [0, 65534]");
            return BWAMS_ERR_UNSUPPORTED;
        }
        if (l > mx) mx = (int)l;
    }
    BWAMS_HIP(hipSetDevice(b->idx->device));
    b->nseq = nseq;
    b->nbases = nb;
    b->max_read_len = mx;
    b->has_skip = skip != nullptr;
    b->seed_done = false;
    // enc may already live in this GPU's memory (a caller that keeps several chunks resident): the copy kind is inferred
    if (nb) BWAMS_HIP(hipMemcpyAsync(b->d_enc, enc, (size_t)nb, hipMemcpyDefault, b->stream));
    BWAMS_HIP(hipMemcpyAsync(b->d_cum, cum, (size_t)(nseq + 1) * 8, hipMemcpyHostToDevice, b->stream));
    if (skip && nseq) BWAMS_HIP(hipMemcpyAsync(b->d_skip, skip, (size_t)nseq, hipMemcpyHostToDevice, b->stream));
    // the source buffers belong to the caller: do not return before they are consumed
    BWAMS_HIP(hipStreamSynchronize(b->stream));

    // packed form of the reads: 16 bases per code word + 32 bases per N-mask word, padded to 4 words
    {
        const int cw = (mx + 15) / 16, mw = (mx + 31) / 32;
        int W = ((cw + mw + 3) / 4) * 4;
        if (W < 4) W = 4;
        b->read_w = W;
        b->read_cw = cw;
        const int64_t need = (int64_t)W * (nseq > 0 ? nseq : 1);
        if (need > b->packed_cap) {
            if (b->d_packed) (void)hipFree(b->d_packed);
            b->d_packed = nullptr;
            BWAMS_HIP(dev_malloc(&b->d_packed, (size_t)need * 4));
            b->packed_cap = need;
        }
    }
    // per-lane scratch for the previous-interval lists: (longest read + 1) entries per lane
    const int cap = mx + 1;
    const int64_t threads = seed_max_threads(b->cu_count);
    if (cap > b->prev_cap || threads > b->prev_threads) {
        if (b->d_prev) (void)hipFree(b->d_prev);
        b->d_prev = nullptr;
        const size_t n = (size_t)cap * (size_t)threads;
        BWAMS_HIP(dev_malloc(&b->d_prev, n * 16));
        b->prev_cap = cap;
        b->prev_threads = threads;
    }
    // backward phases with long interval lists (smem_bwd_wave_kernel): a slot per read and eight list entries per read cover
    // what uniform and repeat-rich genomes produce several times over; when they are full a pivot simply stays on its lane
    // (two item arrays of bi slots in one allocation: long lists, short lists; once a launch drains every backward phase leaves its
    // lane, about half a pivot per read in flight, profiles/r04_notes.md)
    const int64_t cap_mul = knobs().bwd_cap_mul;                // lab: room for EVERY backward phase
    const int64_t bi = std::max<int64_t>(nseq, 4096) * 2 * cap_mul, be = std::max<int64_t>(nseq, 4096) * 24 * cap_mul;
    if (knobs().seed_split) {
        // lists: round 1 needs 2 (bases + reads) entries, round 2 (max_len + 2) per work item — room for two items per read; items: six per read
        const int64_t fl = std::max<int64_t>(2 * (nb + nseq) + 64, 2 * std::max<int64_t>(nseq, 4096) * (int64_t)(mx + 2));
        const int64_t fi = 6 * std::max<int64_t>(nseq, 4096) + 4096;
        const int dbl = knobs().seed_split == 2 ? 2 : 1;       // lab: two halves (the forward kernel writes one while the backward kernel reads the other)
        if (dbl > b->split_dbl) { b->fl_cap = 0; b->f_items_cap = 0; b->split_dbl = dbl; }      // (the lab's doubled buffers: allocate again)
        if (fl > b->fl_cap) {
            if (b->d_fl_ent) (void)hipFree(b->d_fl_ent);
            b->d_fl_ent = nullptr; b->fl_cap = 0;
            BWAMS_HIP(dev_malloc(&b->d_fl_ent, (size_t)fl * 16 * dbl));
            b->fl_cap = fl;
        }
        if (fi > b->f_items_cap) {
            if (b->d_f_items) (void)hipFree(b->d_f_items);
            b->d_f_items = nullptr; b->f_items_cap = 0;
            BWAMS_HIP(dev_malloc(&b->d_f_items, (size_t)fi * sizeof(BwdItem) * dbl));
            b->f_items_cap = fi;
        }
    }
    if (bi > b->bwd_items_cap) {
        if (b->d_bwd_items) (void)hipFree(b->d_bwd_items);
        if (b->d_bwd_ent) (void)hipFree(b->d_bwd_ent);
        b->d_bwd_items = nullptr; b->d_bwd_ent = nullptr; b->bwd_items_cap = b->bwd_ent_cap = 0;
        BWAMS_HIP(dev_malloc(&b->d_bwd_items, (size_t)bi * 2 * sizeof(BwdItem)));
        BWAMS_HIP(dev_malloc(&b->d_bwd_ent, (size_t)be * 16));
        b->bwd_items_cap = bi;
        b->bwd_ent_cap = be;
    }
    return BWAMS_OK;
}

static int seed_run_once(bwams_batch_t *b, const bwams_seed_opt_t *opt, int with_sa);

// The SMEM and SA buffers grow on demand: the kernels keep counting when a buffer is full, so one
// overflowing pass tells the size the chunk needs and the stage is simply run again.
int bwams_seed_run(bwams_batch_t *b, const bwams_seed_opt_t *opt, int with_sa) {
    if (!b || !opt) return BWAMS_ERR_ARG;
    b->last_seed_opt = *opt;
    b->seed_ert = nullptr;
    int rc = seed_run_once(b, opt, with_sa);
    if (rc == BWAMS_ERR_CAPACITY && (b->n_smem > b->max_smem || b->n_pool_slots > b->pool_cap)) {
        BWAMS_HIP(hipStreamSynchronize(b->stream));
        // the slots handed out (holes included) bound what the chunk needs whatever filled the pool
        const int64_t seen = std::max(b->n_smem, b->n_pool_slots - seed_pool_slack(b->cu_count));
        const int64_t need = std::max(seen, b->max_smem) + seen / 4 + 1024;
        if ((rc = alloc_smem_buffers(b, need))) return rc;
        b->tmp_bytes = 0;                           // rocPRIM scratch is re-queried per call
        if (b->d_tmp) { (void)hipFree(b->d_tmp); b->d_tmp = nullptr; }
        rc = seed_run_once(b, opt, with_sa);
    }
    return rc;
}

static int seed_run_once(bwams_batch_t *b, const bwams_seed_opt_t *opt, int with_sa) {
    if (!b || !opt) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    b->with_sa = with_sa != 0;
    b->n_smem = b->n_sa = 0;

    if (knobs().cp2 && b->idx->cp2_kind != knobs().cp2 && b->idx->d_cp && b->idx->n_blk > 0) {     // the search kernels' own table, once per index
        if (b->idx->d_cp2) { BWAMS_HIP(hipStreamSynchronize(st)); (void)hipFree(b->idx->d_cp2); b->idx->d_cp2 = nullptr; }
        BWAMS_HIP(dev_malloc(&b->idx->d_cp2, cp2_bytes(b->idx->n_blk, knobs().cp2)));
        launch_cp2_build(reinterpret_cast<const uint4 *>(b->idx->d_cp), b->idx->n_blk, reinterpret_cast<uint4 *>(b->idx->d_cp2), knobs().cp2, st);
        b->idx->cp2_kind = knobs().cp2;
    }
    SeedLaunch a;
    a.fmi = b->idx->fmi;
    a.fmi.cp2 = knobs().cp2 ? reinterpret_cast<const uint4 *>(b->idx->d_cp2) : nullptr;
    a.fmi.tab_kind = knobs().cp2 ? b->idx->cp2_kind : 0;
    a.enc = b->d_enc;
    a.cum = b->d_cum;
    a.skip = b->has_skip ? b->d_skip : nullptr;
    a.nseq = b->nseq;
    a.packed = b->d_packed;
    a.read_w = b->read_w;
    a.read_cw = b->read_cw;
    a.reads_in_lds = b->read_w <= 40;      // 40 words x 256 lanes x 4 B = 40 KB per workgroup
    a.debug = knobs().debug;
    a.min_seed_len = opt->min_seed_len;
    a.pool = b->d_pool;
    a.pool_cap = b->pool_cap;
    a.ctr = b->d_ctr;
    a.prev = b->d_prev;
    a.prev_cap = b->prev_cap;
    a.prev_threads = b->prev_threads;
    const bool split = knobs().seed_split && !b->seed_split_failed && b->d_fl_ent && b->d_f_items;
    const bool lab_overlap = split && knobs().seed_split == 2;
    if (lab_overlap) b->split_parity ^= 1;
    const int par = lab_overlap ? b->split_parity : 0;
    a.f_items = b->d_f_items + (int64_t)par * b->f_items_cap;
    a.f_items_cap = b->f_items_cap;
    a.f_items_fixed = -1;
    a.fl_ent = b->d_fl_ent + (int64_t)par * b->fl_cap;
    a.fl_cap = b->fl_cap;
    a.fl_item_stride = b->max_read_len + 2;
    a.bwd_items = b->d_bwd_items;
    a.bwd_items_s = b->d_bwd_items + b->bwd_items_cap;
    a.bwd_ent = b->d_bwd_ent;
    a.bwd_items_cap = b->bwd_items_cap;
    a.bwd_ent_cap = b->bwd_ent_cap;
    {   // hand-over thresholds (fmi_seed.hip, bwd_hand_over; profiles/r03_notes.md 86): 40 entries at the forward end, or 8 still alive
        // after 24 columns; BWAMS_BWD_MIN_LIST=0: every backward phase stays on its lane
        const Knobs &kn = knobs();                           // the tests lower them so that toy genomes reach the kernels behind the search
        a.bwd_min_list = kn.bwd_min_list;
        a.bwd_cols = kn.bwd_cols;
        a.bwd_late_list = kn.bwd_late_list;
        // once the work queue has run dry: 24 entries at the forward end, or 12 alive after 8 columns (profiles/r03_notes.md 96)
        a.bwd_dry_min_list = kn.bwd_dry_min_list;
        a.bwd_dry_cols = kn.bwd_dry_cols;
        a.bwd_dry_late_list = kn.bwd_dry_late_list;
    }
    const int split_len = (int)(opt->min_seed_len * opt->split_factor + .499);

    // events: 0 start | 8,9 round-1 kernel | 10,11 round-2 kernel | 12,13 round-3 kernel | 3 rounds done
    BWAMS_HIP(hipMemsetAsync(b->d_ctr, 0, sizeof(DevCounters), st));
    BWAMS_HIP(hipEventRecord(b->ev[0], st));
    launch_pack_reads(b->d_enc, b->d_cum, b->nseq, b->read_w, b->read_cw, b->d_packed, st);
    launch_mark(b->d_ctr, 0, st);
    // Round 3 reads nothing of rounds 1 and 2 (bwtSeedStrategyAllPosOneThread walks every read from position 0).  BWAMS_SEED_R3_BESIDE=2
    // (an experiment, kept behind the switch and under the parity test): it is launched HERE, on the stream of its own, with a pool and
    // counters of its own (round 2's work list is cut from the main pool's prefix); behind round 2 its records are appended to the main
    // pool and its counts folded in (append_r3_kernel + mark 3).  The hope was that its workgroups would only find room where round 1
    // drains; they are placed beside round 1's throughout: round 1 16.1 -> 18.3 ms, round 2 12.5 -> 10.7, the stage 35.3 -> 35.8 ms.
    const int r3_mode = (b->nseq > 0 && opt->max_mem_intv > 0 && !split && !lab_overlap) ? knobs().r3_beside : 0;
    SeedLaunch a3 = a;
    a3.min_seed_len = opt->min_seed_len + 1;
    if (r3_mode == 2) {
        if (!b->d_pool3) BWAMS_HIP(dev_malloc(&b->d_pool3, (size_t)b->pool3_cap * sizeof(bwams_smem_t)));
        a3.pool = b->d_pool3; a3.pool_cap = b->pool3_cap; a3.ctr = b->d_ctr3;
        BWAMS_HIP(hipMemsetAsync(b->d_ctr3, 0, sizeof(DevCounters), st));
        BWAMS_HIP(hipEventRecord(b->seed_fork, st));
        BWAMS_HIP(hipStreamWaitEvent(b->seed_aux, b->seed_fork, 0));
        BWAMS_HIP(hipEventRecord(b->ev[12], b->seed_aux));
        launch_smem_round3(a3, opt->max_mem_intv, b->cu_count, b->seed_aux);
        BWAMS_HIP(hipEventRecord(b->ev[13], b->seed_aux));
        BWAMS_HIP(hipEventRecord(b->seed_join, b->seed_aux));
    }
    BWAMS_HIP(hipEventRecord(b->ev[8], st));
    if (b->nseq > 0) {
        if (lab_overlap && b->f_items_prev[par ^ 1][0] >= 0) {
            // LAB ONLY: the backward kernel over the items the PREVIOUS run left in the other half (the same reads: the same SMEMs), beside
            // this run's forward kernel — what a perfect overlap of the two would take
            SeedLaunch ab = a;
            ab.f_items = b->d_f_items + (int64_t)(par ^ 1) * b->f_items_cap;
            ab.fl_ent = b->d_fl_ent + (int64_t)(par ^ 1) * b->fl_cap;
            ab.f_items_fixed = b->f_items_prev[par ^ 1][0];
            BWAMS_HIP(hipEventRecord(b->seed_fork, st));
            BWAMS_HIP(hipStreamWaitEvent(b->seed_aux, b->seed_fork, 0));
            launch_smem_fwd(a, nullptr, b->cu_count, st);
            launch_smem_bwdl(ab, b->cu_count, b->seed_aux);
            BWAMS_HIP(hipEventRecord(b->seed_join, b->seed_aux));
            BWAMS_HIP(hipStreamWaitEvent(st, b->seed_join, 0));
        } else if (split) { launch_smem_fwd(a, nullptr, b->cu_count, st); launch_smem_bwdl(a, b->cu_count, st); }
        else launch_smem_round1(a, b->cu_count, st);
    }
#ifdef BWAMS_BWDDBG
    static hipEvent_t dbg_ev = nullptr;
    if (!dbg_ev) BWAMS_HIP(hipEventCreate(&dbg_ev));
    BWAMS_HIP(hipEventRecord(dbg_ev, st));
#endif
    if (b->nseq > 0) launch_smem_bwd_wave(a, b->cu_count, st);
    BWAMS_HIP(hipEventRecord(b->ev[9], st));
    launch_mark(b->d_ctr, 1, st);
    if (b->nseq > 0) launch_round2_work(a, b->d_work2, b->pool_cap, split_len, opt->split_width, b->cu_count, st);
    // Round 3 reads nothing of rounds 1 and 2 (bwtSeedStrategyAllPosOneThread walks every read from position 0): it runs beside
    // round 2 on a stream of its own and fills the tail in which round 2's slowest reads keep few lanes busy.  Its extensions and
    // SMEMs are counted apart (n_ext3 / n_blk3 / n_smem3), so that the per-round figures stay exact.
    const bool r3_beside = knobs().r3_beside != 0 && r3_mode != 2;
    const bool r3 = b->nseq > 0 && opt->max_mem_intv > 0 && r3_mode != 2;
    hipStream_t st3 = r3_beside ? b->seed_aux : st;
    if (r3 && r3_beside) {
        BWAMS_HIP(hipEventRecord(b->seed_fork, st));
        BWAMS_HIP(hipStreamWaitEvent(st3, b->seed_fork, 0));
        BWAMS_HIP(hipEventRecord(b->ev[12], st3));
        launch_smem_round3(a3, opt->max_mem_intv, b->cu_count, st3);
        BWAMS_HIP(hipEventRecord(b->ev[13], st3));
        BWAMS_HIP(hipEventRecord(b->seed_join, st3));
    }
    BWAMS_HIP(hipEventRecord(b->ev[10], st));
    if (b->nseq > 0) {
        if (split && !lab_overlap) { launch_smem_fwd(a, b->d_work2, b->cu_count, st); launch_smem_bwdl(a, b->cu_count, st); }
        else launch_smem_round2(a, b->d_work2, b->cu_count, st);       // (the lab's overlap run keeps round 1's items intact for the next run)
    }
    if (b->nseq > 0) launch_smem_bwd_wave(a, b->cu_count, st);
    BWAMS_HIP(hipEventRecord(b->ev[11], st));
    if ((r3 && r3_beside) || r3_mode == 2) BWAMS_HIP(hipStreamWaitEvent(st, b->seed_join, 0));
    launch_mark(b->d_ctr, 2, st);
    if (r3_mode == 2) launch_append_r3(b->d_pool, b->pool_cap, b->d_pool3, b->pool3_cap, b->d_ctr, b->d_ctr3, st);
    if (!(r3 && r3_beside) && r3_mode != 2) {
        BWAMS_HIP(hipEventRecord(b->ev[12], st));
        if (r3) launch_smem_round3(a3, opt->max_mem_intv, b->cu_count, st);
        BWAMS_HIP(hipEventRecord(b->ev[13], st));
    }
    launch_mark(b->d_ctr, 3, st);
    BWAMS_HIP(hipEventRecord(b->ev[3], st));
    BWAMS_HIP(hipGetLastError());
    // the SMEM count sizes the sort: one small read-back
    BWAMS_HIP(hipMemcpyAsync(b->h_ctr, b->d_ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
#ifdef BWAMS_BWDDBG
    if (knobs().verbose) {
        const unsigned long long *d = b->h_ctr->dbg;
        { float m1 = 0, m2 = 0; (void)hipEventElapsedTime(&m1, b->ev[8], dbg_ev); (void)hipEventElapsedTime(&m2, dbg_ev, b->ev[9]);
          fprintf(stderr, "[smem_r1] search kernel %.3f ms, the two backward kernels behind it %.3f ms\n", m1, m2); }
        fprintf(stderr, "[bwd_wave] rounds 1+2: items %llu, column batches %llu (%.1f per item), waves with work %llu: busy mean %.3f ms max %.3f ms, "
                "of it between items (ticket, item, list, read) %.1f %%, per column batch %.2f us\n", d[0], d[1], d[0] ? (double)d[1] / d[0] : 0.0, d[5],
                d[5] ? d[2] / (double)d[5] * 1e-5 : 0.0, d[4] * 1e-5, d[2] ? 100.0 * d[3] / d[2] : 0.0, d[1] ? (d[2] - d[3]) * 1e-2 / d[1] : 0.0);
        fprintf(stderr, "[bwd_group] rounds 1+2: items %llu, wave-iterations %llu (groups live per iteration %.2f), waves with work %llu: busy mean %.3f ms max %.3f ms; "
                "first in %.3f last out %.3f ms after round 1's start; [bwd_wave] first in %.3f last out %.3f\n", d[68], d[69], d[69] ? (double)d[70] / d[69] : 0.0, d[72],
                d[72] ? d[71] * 1e-5 / d[72] : 0.0, d[73] * 1e-5, (~d[75] - ~d[8]) * 1e-5, (d[74] - ~d[8]) * 1e-5, (~d[7] - ~d[8]) * 1e-5, (d[6] - ~d[8]) * 1e-5);
        fprintf(stderr, "[bwd_group] extensions %llu of %llu (rounds 1+2)\n", d[76], (unsigned long long)b->h_ctr->ext_after[1]);
        const unsigned long long t0 = ~d[8], tdry = ~d[9];
        fprintf(stderr, "[smem_r1] waves %llu: read queue dry at %.3f ms, last wave out at %.3f ms, mean wave life %.3f ms (%.3f ms of it after the queue ran dry); "
                "iterations %llu, lanes extending per iteration %.1f\n", d[12], (tdry - t0) * 1e-5, (d[10] - t0) * 1e-5, d[12] ? d[11] * 1e-5 / d[12] : 0.0,
                d[12] ? d[15] * 1e-5 / d[12] : 0.0, d[13], d[13] ? (double)d[14] / d[13] : 0.0);
        fprintf(stderr, "[smem_r1] waves leaving per 0.4 ms:");
        for (int i = 0; i < 48; ++i) if (d[16 + i]) fprintf(stderr, " %.1f:%llu", i * 0.4, d[16 + i]);
        fprintf(stderr, "\n[smem_r1] wave-iterations after the wave first saw the queue dry: %llu, with one lane extending %llu (max per wave %llu), with 2-4 lanes %llu\n", d[64], d[65], d[67], d[66]);
    }
#endif
    if (lab_overlap) { b->f_items_prev[par][0] = (int64_t)b->h_ctr->f_items_r[0]; b->f_items_prev[par][1] = (int64_t)b->h_ctr->f_items_r[1]; }
    if (knobs().verbose && split) fprintf(stderr, "[bwams_seed_run] split search: items r1 %llu r2 %llu, overflow %llu%s\n", b->h_ctr->f_items_r[0], b->h_ctr->f_items_r[1],
                                          b->h_ctr->f_overflow, lab_overlap ? " (lab: backward kernel beside the forward kernel)" : "");
    if (split && b->h_ctr->f_overflow) {         // pivots that found no room between the two kernels: this batch searches unsplit from now on
        b->seed_split_failed = true;
        return seed_run_once(b, opt, with_sa);
    }
    const int64_t n_slots = (int64_t)b->h_ctr->n_smem_total;      // pool slots handed out (holes included)
    const int64_t n = (int64_t)b->h_ctr->n_smem_valid;           // real SMEMs
    b->n_smem = n;
    b->n_pool_slots = n_slots;
    if (n > b->max_smem || n_slots > b->pool_cap) {
        set_last_error("SMEM pool overflow: need " + std::to_string(n) + " slots");
        b->seed_done = true;
        return BWAMS_ERR_CAPACITY;
    }
    if (n_slots > 0) {
        // key = rid << 32 | m << 16 | n; chunk holes carry rid = nseq and sort behind every read
        launch_make_keys(b->d_pool, n_slots, b->d_keys, b->d_vals, (uint32_t)b->nseq, st);
        int rid_bits = 1;
        while (((int64_t)1 << rid_bits) <= b->nseq) rid_bits++;
        size_t tb = 0;      // the temporary size depends on the size / bit range: ask for this call
        BWAMS_HIP(rocprim::radix_sort_pairs(nullptr, tb, b->d_keys, b->d_keys2, b->d_vals, b->d_vals2,
                                            (size_t)n_slots, 0, 32 + rid_bits, st));
        if (tb > b->tmp_bytes) {
            BWAMS_HIP(hipStreamSynchronize(st));
            (void)hipFree(b->d_tmp);
            b->d_tmp = nullptr;
            BWAMS_HIP(dev_malloc(&b->d_tmp, tb));
            b->tmp_bytes = tb;
        }
        tb = b->tmp_bytes;
        BWAMS_HIP(rocprim::radix_sort_pairs(b->d_tmp, tb, b->d_keys, b->d_keys2, b->d_vals, b->d_vals2,
                                            (size_t)n_slots, 0, 32 + rid_bits, st));
        launch_gather_sorted(b->d_pool, b->d_vals2, n, b->d_sorted, with_sa ? b->d_sa_cnt : nullptr,
                             opt->max_occ, st);
    }
    BWAMS_HIP(hipEventRecord(b->ev[4], st));
    if (with_sa && n > 0) {
        size_t tb = 0;
        BWAMS_HIP(rocprim::exclusive_scan(nullptr, tb, b->d_sa_cnt, b->d_sa_off, (int64_t)0, (size_t)n + 1,
                                          rocprim::plus<int64_t>(), st));
        if (tb > b->tmp_bytes) {
            BWAMS_HIP(hipStreamSynchronize(st));
            (void)hipFree(b->d_tmp);
            b->d_tmp = nullptr;
            BWAMS_HIP(dev_malloc(&b->d_tmp, tb));
            b->tmp_bytes = tb;
        }
        tb = b->tmp_bytes;
        BWAMS_HIP(hipMemsetAsync(b->d_sa_cnt + n, 0, 8, st));
        BWAMS_HIP(rocprim::exclusive_scan(b->d_tmp, tb, b->d_sa_cnt, b->d_sa_off, (int64_t)0, (size_t)n + 1,
                                          rocprim::plus<int64_t>(), st));
        launch_sa_lookup(b->idx->fmi, b->d_sorted, n, b->d_sa_off, b->d_sa_coord, b->max_sa, opt->max_occ,
                         b->d_ctr, b->cu_count, st);
    }
    BWAMS_HIP(hipEventRecord(b->ev[5], st));
    BWAMS_HIP(hipGetLastError());
    b->seed_done = true;
    return BWAMS_OK;
}

static int ert_redo_ensure(bwams_batch_t *b, int64_t n) {
    if (n <= b->cap_ert_redo) return BWAMS_OK;
    if (b->d_ert_redo) (void)hipFree(b->d_ert_redo);
    b->d_ert_redo = nullptr;
    b->cap_ert_redo = n + n / 4 + 1024;
    BWAMS_HIP(dev_malloc(&b->d_ert_redo, (size_t)((b->cap_ert_redo + 31) / 32) * 4));
    return BWAMS_OK;
}

int bwams_seed_counts(bwams_batch_t *b, int64_t *n_smem, int64_t *n_sa) {
    if (!b || !b->seed_done) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    BWAMS_HIP(hipMemcpyAsync(b->h_ctr, b->d_ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, b->stream));
    BWAMS_HIP(hipStreamSynchronize(b->stream));
    b->n_sa = b->with_sa ? (int64_t)b->h_ctr->n_sa_lookups : 0;
    if (n_smem) *n_smem = b->n_smem;
    if (n_sa) *n_sa = b->n_sa;
    if (b->n_smem > b->max_smem) return BWAMS_ERR_CAPACITY;
    if (b->n_sa > b->max_sa) {
        // the lookup kernel counted every coordinate but stored only max_sa of them: grow and run it again
        (void)hipFree(b->d_sa_coord);
        b->d_sa_coord = nullptr;
        b->max_sa = b->n_sa + b->n_sa / 8 + 1024;
        BWAMS_HIP(dev_malloc(&b->d_sa_coord, (size_t)b->max_sa * 8));
        BWAMS_HIP(hipMemsetAsync(&b->d_ctr->n_sa_lookups, 0, 2 * sizeof(unsigned long long), b->stream));   // + n_lf_steps
        if (b->seed_ert) {
            launch_ert_locate(b->seed_ert->t, b->d_enc, b->d_cum, b->d_sorted, b->n_smem, b->d_sa_cnt, b->last_seed_opt.max_occ,
                              b->d_ctr, b->d_ert_stk, b->ert_stk_frames, b->cu_count, b->stream);
            if (int rrc = ert_redo_ensure(b, b->n_smem)) return rrc;
            launch_ert_gather(b->seed_ert->t, b->d_sorted, b->n_smem, b->d_sa_off, b->d_sa_coord, b->max_sa,
                              b->last_seed_opt.max_occ, b->d_ctr, b->d_ert_stk, b->ert_stk_frames, b->d_ert_redo, b->max_sa,
                              b->cu_count, b->stream);
            launch_ert_clear(b->d_sorted, b->n_smem, b->stream);
        } else
        launch_sa_lookup(b->idx->fmi, b->d_sorted, b->n_smem, b->d_sa_off, b->d_sa_coord, b->max_sa, b->last_seed_opt.max_occ,
                         b->d_ctr, b->cu_count, b->stream);
        BWAMS_HIP(hipEventRecord(b->ev[5], b->stream));
        BWAMS_HIP(hipMemcpyAsync(b->h_ctr, b->d_ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, b->stream));
        BWAMS_HIP(hipStreamSynchronize(b->stream));
        b->n_sa = (int64_t)b->h_ctr->n_sa_lookups;
        if (n_sa) *n_sa = b->n_sa;
        if (b->n_sa > b->max_sa) {
            set_last_error("SA coordinate buffer overflow: need " + std::to_string(b->n_sa));
            return BWAMS_ERR_CAPACITY;
        }
    }
    return BWAMS_OK;
}

int bwams_seed_fetch(bwams_batch_t *b, bwams_smem_t *smem_out, int64_t smem_cap, int64_t *sa_coord,
                     int64_t sa_cap, int64_t *sa_off) {
    if (!b || !b->seed_done) return BWAMS_ERR_ARG;
    int64_t ns = 0, na = 0;
    int rc = bwams_seed_counts(b, &ns, &na);
    if (rc) return rc;
    if (ns > smem_cap || (sa_coord && na > sa_cap)) {
        set_last_error("bwams_seed_fetch: caller buffers too small");
        return BWAMS_ERR_CAPACITY;
    }
    if (smem_out && ns)
        BWAMS_HIP(hipMemcpyAsync(smem_out, b->d_sorted, (size_t)ns * sizeof(bwams_smem_t), hipMemcpyDeviceToHost,
                                 b->stream));
    if (sa_coord && sa_off && b->with_sa) {
        if (ns) {
            BWAMS_HIP(hipMemcpyAsync(sa_off, b->d_sa_off, (size_t)(ns + 1) * 8, hipMemcpyDeviceToHost, b->stream));
            if (na)
                BWAMS_HIP(hipMemcpyAsync(sa_coord, b->d_sa_coord, (size_t)na * 8, hipMemcpyDeviceToHost, b->stream));
        } else {
            sa_off[0] = 0;
        }
    }
    BWAMS_HIP(hipStreamSynchronize(b->stream));
    return BWAMS_OK;
}

int bwams_seed_fmi(bwams_batch_t *b, const uint8_t *enc, const int64_t *cum, const uint8_t *skip, int64_t nseq,
                   const bwams_seed_opt_t *opt, bwams_smem_t *smem_out, int64_t smem_cap, int64_t *n_smem,
                   int64_t *sa_coord, int64_t sa_cap, int64_t *sa_off, int64_t *n_sa) {
    int rc = bwams_seed_upload(b, enc, cum, skip, nseq);
    if (rc) return rc;
    const int with_sa = sa_coord && sa_off;
    rc = bwams_seed_run(b, opt, with_sa);
    if (rc) {
        if (n_smem) *n_smem = b->n_smem;
        return rc;
    }
    rc = bwams_seed_counts(b, n_smem, n_sa);
    if (rc) return rc;
    return bwams_seed_fetch(b, smem_out, smem_cap, sa_coord, sa_cap, sa_off);
}

/* ------------------------------------------------------------ ERT seeding -- */

// A loaded index starts with an empty hit-count table (filled as big subtrees are counted for the first time); sized
// from the tree bytes: a four-way node with 20 or more hits below it stands for at least some hundred bytes of trees.
// the resident entry + tree-head table (DevErt::fat): 64 bytes per k-mer, derived from the two tables once per index
static int ert_fat_table(bwams_ert *e) {
    if (!knobs().ert_fat) return BWAMS_OK;
    const size_t bytes = (size_t)64 << (2 * e->t.K);
    BWAMS_HIP(dev_malloc(&e->d_fat, bytes));
    launch_ert_fat(e->t, e->mlt_bytes, (uint8_t *)e->d_fat, 0);
    BWAMS_HIP(hipDeviceSynchronize());
    BWAMS_HIP(hipGetLastError());
    e->t.fat = (const uint8_t *)e->d_fat;
    e->bytes += (int64_t)bytes;
    return BWAMS_OK;
}
static int ert_count_table(bwams_ert *e) {
    int bits = 16;
    while (bits < 26 && ((int64_t)1 << bits) < e->mlt_bytes / 256) bits++;
    BWAMS_HIP(dev_malloc(&e->d_cnt, (size_t)16 << bits));
    BWAMS_HIP(hipMemset(e->d_cnt, 0, (size_t)16 << bits));
    e->t.cnt_tab = (uint64_t *)e->d_cnt;
    e->t.cnt_bits = bits;
    e->bytes += (int64_t)16 << bits;
    return BWAMS_OK;
}

int bwams_ert_from_host(bwams_index_t *ix, const uint64_t *kmer_table, int32_t kmer_size, int32_t xmer_size,
                        int32_t read_len, const uint8_t *mlt_table, int64_t mlt_bytes, bwams_ert_t **out) {
    if (!ix || !out || !kmer_table || (mlt_bytes && !mlt_table) || mlt_bytes < 0) return BWAMS_ERR_ARG;
    if (kmer_size < 2 || kmer_size > 15 || xmer_size < 1 || xmer_size > 8 || read_len < kmer_size + xmer_size) {
        set_last_error("bwams_ert_from_host: k-mer size must be in [2, 15], x-mer size in [1, 8]");
        return BWAMS_ERR_ARG;
    }
    if (!ix->d_ref) {
        set_last_error("bwams_ert_from_host: the index was opened without its .0123 reference");
        return BWAMS_ERR_ARG;
    }
    BWAMS_HIP(hipSetDevice(ix->device));
    bwams_ert *e = new bwams_ert();
    e->idx = ix;
    const size_t nk = (size_t)1 << (2 * kmer_size);
    hipError_t he = dev_malloc(&e->d_kmer, nk * 8);
    if (he == hipSuccess) he = dev_malloc(&e->d_mlt, (size_t)mlt_bytes + 16);
    if (he == hipSuccess) he = hipMemcpy(e->d_kmer, kmer_table, nk * 8, hipMemcpyHostToDevice);
    if (he == hipSuccess && mlt_bytes) he = hipMemcpy(e->d_mlt, mlt_table, (size_t)mlt_bytes, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemset((uint8_t *)e->d_mlt + mlt_bytes, 0, 16);
    if (he != hipSuccess) {
        set_last_error(std::string("bwams_ert_from_host: ") + hipGetErrorString(he));
        bwams_ert_close(e);
        return he == hipErrorOutOfMemory ? BWAMS_ERR_NOMEM : BWAMS_ERR_DEVICE;
    }
    e->t.kmer = (const uint64_t *)e->d_kmer;
    e->t.mlt = (const uint8_t *)e->d_mlt;
    e->t.ref = ix->fmi.ref;
    e->t.ref_len = ix->fmi.ref_seq_len - 1;
    e->t.K = kmer_size; e->t.X = xmer_size; e->t.read_len = read_len;
    e->bytes = (int64_t)(nk * 8) + mlt_bytes + 16;
    e->mlt_bytes = mlt_bytes;
    if (int crc = ert_count_table(e)) { bwams_ert_close(e); return crc; }
    if (int crc = ert_fat_table(e)) { bwams_ert_close(e); return crc; }
    *out = e;
    return BWAMS_OK;
}

int bwams_ert_open(bwams_index_t *ix, const char *prefix, int32_t read_len, bwams_ert_t **out) {
    if (!ix || !prefix || !out) return BWAMS_ERR_ARG;
    if (!ix->d_ref) {
        set_last_error("bwams_ert_open: the index was opened without its .0123 reference");
        return BWAMS_ERR_ARG;
    }
    const int K = 15, X = 4;                       // kmerSize / xmerSize, src/macro.h:204-206
    const std::string fk = std::string(prefix) + ".kmer_table", fm = std::string(prefix) + ".mlt_table";
    FILE *f1 = fopen(fk.c_str(), "rb"), *f2 = fopen(fm.c_str(), "rb");
    if (!f1 || !f2) {
        if (f1) fclose(f1);
        if (f2) fclose(f2);
        set_last_error("bwams_ert_open: cannot open " + (f1 ? fm : fk));
        return BWAMS_ERR_IO;
    }
    fseek(f2, 0, SEEK_END);
    const int64_t mlt_bytes = (int64_t)ftell(f2);
    fseek(f2, 0, SEEK_SET);
    BWAMS_HIP(hipSetDevice(ix->device));
    bwams_ert *e = new bwams_ert();
    e->idx = ix;
    const size_t nk = (size_t)1 << (2 * K);
    int rc = BWAMS_OK;
    const size_t chunk = (size_t)256 << 20;          // streamed through one pinned staging buffer
    void *stage = nullptr;
    hipError_t he = dev_malloc(&e->d_kmer, nk * 8);
    if (he == hipSuccess) he = dev_malloc(&e->d_mlt, (size_t)mlt_bytes + 16);
    if (he == hipSuccess) he = hipHostMalloc(&stage, chunk);
    if (he != hipSuccess) rc = he == hipErrorOutOfMemory ? BWAMS_ERR_NOMEM : BWAMS_ERR_DEVICE;
    auto stream_in = [&](FILE *f, void *dst, size_t total) {
        size_t done = 0;
        while (rc == BWAMS_OK && done < total) {
            const size_t n = total - done < chunk ? total - done : chunk;
            if (fread(stage, 1, n, f) != n) { rc = BWAMS_ERR_IO; break; }
            if (hipMemcpy((uint8_t *)dst + done, stage, n, hipMemcpyHostToDevice) != hipSuccess) { rc = BWAMS_ERR_DEVICE; break; }
            done += n;
        }
    };
    if (rc == BWAMS_OK) stream_in(f1, e->d_kmer, nk * 8);
    if (rc == BWAMS_OK) stream_in(f2, e->d_mlt, (size_t)mlt_bytes);
    if (rc == BWAMS_OK && hipMemset((uint8_t *)e->d_mlt + mlt_bytes, 0, 16) != hipSuccess) rc = BWAMS_ERR_DEVICE;
    fclose(f1); fclose(f2);
    if (stage) (void)hipHostFree(stage);
    if (rc != BWAMS_OK) {
        set_last_error("bwams_ert_open: reading " + fk + " / " + fm + " failed");
        bwams_ert_close(e);
        return rc;
    }
    e->t.kmer = (const uint64_t *)e->d_kmer;
    e->t.mlt = (const uint8_t *)e->d_mlt;
    e->t.ref = ix->fmi.ref;
    e->t.ref_len = ix->fmi.ref_seq_len - 1;
    e->t.K = K; e->t.X = X; e->t.read_len = read_len;
    e->bytes = (int64_t)(nk * 8) + mlt_bytes + 16;
    e->mlt_bytes = mlt_bytes;
    if (int crc = ert_count_table(e)) { bwams_ert_close(e); return crc; }
    if (int crc = ert_fat_table(e)) { bwams_ert_close(e); return crc; }
    *out = e;
    return BWAMS_OK;
}

int bwams_ert_build(bwams_index_t *ix, int32_t kmer_size, int32_t xmer_size, int32_t read_len, int32_t hit_threshold,
                    bwams_ert_t **out) {
    if (!ix || !out) return BWAMS_ERR_ARG;
    if (kmer_size < 2 || kmer_size > 15 || xmer_size < 1 || xmer_size > 8 || read_len < kmer_size + xmer_size || read_len > 255 ||
        hit_threshold < 1) {
        set_last_error("bwams_ert_build: k-mer size must be in [2, 15], x-mer size in [1, 8], read length in [k + x, 255]");
        return BWAMS_ERR_ARG;
    }
    if (!ix->d_ref) {
        set_last_error("bwams_ert_build: the index holds no .0123 reference (leaf expansion reads it)");
        return BWAMS_ERR_ARG;
    }
    BWAMS_HIP(hipSetDevice(ix->device));
    hipDeviceProp_t prop;
    BWAMS_HIP(hipGetDeviceProperties(&prop, ix->device));
    bwams_ert *e = new bwams_ert();
    e->idx = ix;
    const int rc = ert_build_device(e, ix->fmi, kmer_size, xmer_size, read_len, hit_threshold, prop.multiProcessorCount,
                                    knobs().verbose != 0);
    if (rc) { bwams_ert_close(e); return rc; }
    if (int crc = ert_fat_table(e)) { bwams_ert_close(e); return crc; }
    *out = e;
    return BWAMS_OK;
}

int bwams_ert_info(const bwams_ert_t *e, int32_t *kmer_size, int32_t *xmer_size, int32_t *read_len, int64_t *mlt_bytes,
                   float build_ms[3]) {
    if (!e) return BWAMS_ERR_ARG;
    if (kmer_size) *kmer_size = e->t.K;
    if (xmer_size) *xmer_size = e->t.X;
    if (read_len) *read_len = e->t.read_len;
    if (mlt_bytes) *mlt_bytes = e->mlt_bytes;
    if (build_ms) for (int i = 0; i < 3; ++i) build_ms[i] = e->build_ms[i];
    return BWAMS_OK;
}

int bwams_ert_fetch(bwams_ert_t *e, uint64_t *kmer_table, uint8_t *mlt_table) {
    if (!e) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(e->idx->device));
    if (kmer_table) BWAMS_HIP(hipMemcpy(kmer_table, e->d_kmer, ((size_t)1 << (2 * e->t.K)) * 8, hipMemcpyDeviceToHost));
    if (mlt_table && e->mlt_bytes) BWAMS_HIP(hipMemcpy(mlt_table, e->d_mlt, (size_t)e->mlt_bytes, hipMemcpyDeviceToHost));
    return BWAMS_OK;
}

int bwams_ert_save(bwams_ert_t *e, const char *prefix) {
    if (!e || !prefix) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(e->idx->device));
    const size_t chunk = (size_t)256 << 20;
    void *stage = nullptr;
    BWAMS_HIP(hipHostMalloc(&stage, chunk));
    int rc = BWAMS_OK;
    auto stream_out = [&](const std::string &path, const void *src, size_t total) {
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) { rc = BWAMS_ERR_IO; set_last_error("bwams_ert_save: cannot create " + path); return; }
        size_t done = 0;
        while (rc == BWAMS_OK && done < total) {
            const size_t n = total - done < chunk ? total - done : chunk;
            if (hipMemcpy(stage, (const uint8_t *)src + done, n, hipMemcpyDeviceToHost) != hipSuccess) { rc = BWAMS_ERR_DEVICE; break; }
            if (fwrite(stage, 1, n, f) != n) { rc = BWAMS_ERR_IO; set_last_error("bwams_ert_save: short write to " + path); break; }
            done += n;
        }
        fclose(f);
    };
    stream_out(std::string(prefix) + ".kmer_table", e->d_kmer, ((size_t)1 << (2 * e->t.K)) * 8);
    if (rc == BWAMS_OK) stream_out(std::string(prefix) + ".mlt_table", e->d_mlt, (size_t)e->mlt_bytes);
    (void)hipHostFree(stage);
    return rc;
}

int bwams_ert_close(bwams_ert_t *e) {
    if (!e) return BWAMS_OK;
    (void)hipSetDevice(e->idx->device);
    if (e->d_kmer) (void)hipFree(e->d_kmer);
    if (e->d_mlt) (void)hipFree(e->d_mlt);
    if (e->d_cnt) (void)hipFree(e->d_cnt);
    if (e->d_fat) (void)hipFree(e->d_fat);
    delete e;
    return BWAMS_OK;
}

int64_t bwams_ert_bytes(const bwams_ert_t *e) { return e ? e->bytes : 0; }

int bwams_ert_set_fat(bwams_ert_t *e, int32_t on) {
    if (!e) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(e->idx->device));
    BWAMS_HIP(hipDeviceSynchronize());                   // no walk is reading it
    if (on) {
        if (e->d_fat) return BWAMS_OK;
        const size_t bytes = (size_t)64 << (2 * e->t.K);
        BWAMS_HIP(dev_malloc(&e->d_fat, bytes));
        launch_ert_fat(e->t, e->mlt_bytes, (uint8_t *)e->d_fat, 0);
        BWAMS_HIP(hipDeviceSynchronize());
        e->t.fat = (const uint8_t *)e->d_fat;
        e->bytes += (int64_t)bytes;
        return BWAMS_OK;
    }
    if (e->d_fat) {
        (void)hipFree(e->d_fat);
        e->d_fat = nullptr;
        e->t.fat = nullptr;
        e->bytes -= (int64_t)64 << (2 * e->t.K);
    }
    return BWAMS_OK;
}

static int ert_run_once(bwams_batch_t *b, bwams_ert_t *e, const bwams_seed_opt_t *opt, int with_sa, int M) {
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    b->with_sa = with_sa != 0;
    b->n_smem = b->n_sa = 0;
    const int64_t need = (int64_t)ert_prof_bytes(b->nbases);
    if (need > b->cap_ert_prof) {
        if (b->d_ert_prof) (void)hipFree(b->d_ert_prof);
        b->d_ert_prof = nullptr;
        b->cap_ert_prof = need + need / 8;
        BWAMS_HIP(dev_malloc(&b->d_ert_prof, (size_t)b->cap_ert_prof));
    }
    const int frames = 2 * (e->t.read_len + 2);      // the counting walk keeps two words per level
    const size_t part_bytes = ert_count_bytes();      // partial counters sit behind the stacks
    if (frames > b->ert_stk_frames) {
        if (b->d_ert_stk) (void)hipFree(b->d_ert_stk);
        b->d_ert_stk = nullptr;
        BWAMS_HIP(dev_malloc(&b->d_ert_stk, (size_t)ert_walk_threads(b->cu_count) * (size_t)frames * 8 + part_bytes));
        b->ert_stk_frames = frames;
        BWAMS_HIP(hipMemsetAsync(b->d_ert_stk + (size_t)ert_walk_threads(b->cu_count) * (size_t)frames, 0, part_bytes, b->stream));
    }
    const uint8_t *skip = b->has_skip ? b->d_skip : nullptr;
    // events: 0 start | 8,9 match profiles | 10,11 the three rounds | 3,4 sort | 12,13 locate | 4,5 locate + hits
    BWAMS_HIP(hipMemsetAsync(b->d_ctr, 0, sizeof(DevCounters), st));
    BWAMS_HIP(hipEventRecord(b->ev[0], st));
    BWAMS_HIP(hipEventRecord(b->ev[8], st));
    launch_ert_profile(e->t, b->d_enc, b->d_cum, skip, b->nseq, b->nbases, M, b->d_ert_prof, b->d_ctr,
                       (unsigned long long *)(b->d_ert_stk + (size_t)ert_walk_threads(b->cu_count) * (size_t)b->ert_stk_frames), b->cu_count, st);
    BWAMS_HIP(hipEventRecord(b->ev[9], st));
    BWAMS_HIP(hipEventRecord(b->ev[10], st));
    launch_ert_select(b->d_ert_prof, b->d_cum, skip, b->nseq, b->nbases, M, *opt, b->d_pool, b->pool_cap, b->d_ctr, b->cu_count, st);
    BWAMS_HIP(hipEventRecord(b->ev[11], st));
    BWAMS_HIP(hipEventRecord(b->ev[3], st));
    BWAMS_HIP(hipGetLastError());
    BWAMS_HIP(hipMemcpyAsync(&b->d_ctr->n_smem_valid, &b->d_ctr->n_smem_total, 8, hipMemcpyDeviceToDevice, st));
    for (int k = 0; k < 3; ++k)      // the rounds are not separate launches here: all seeds are reported under round 1
        BWAMS_HIP(hipMemcpyAsync(&b->d_ctr->valid_after[k], &b->d_ctr->n_smem_total, 8, hipMemcpyDeviceToDevice, st));
    BWAMS_HIP(hipMemcpyAsync(b->h_ctr, b->d_ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    const int64_t n = (int64_t)b->h_ctr->n_smem_total;
    b->n_smem = n;
    if (n > b->max_smem || n > b->pool_cap) {
        set_last_error("SMEM pool overflow: need " + std::to_string(n) + " slots");
        b->seed_done = true;
        return BWAMS_ERR_CAPACITY;
    }
    if (n > 0) {
        launch_make_keys(b->d_pool, n, b->d_keys, b->d_vals, (uint32_t)b->nseq, st);
        int rid_bits = 1;
        while (((int64_t)1 << rid_bits) <= b->nseq) rid_bits++;
        size_t tb = 0;
        BWAMS_HIP(rocprim::radix_sort_pairs(nullptr, tb, b->d_keys, b->d_keys2, b->d_vals, b->d_vals2, (size_t)n, 0,
                                            32 + rid_bits, st));
        if (tb > b->tmp_bytes) {
            BWAMS_HIP(hipStreamSynchronize(st));
            (void)hipFree(b->d_tmp);
            b->d_tmp = nullptr;
            BWAMS_HIP(dev_malloc(&b->d_tmp, tb));
            b->tmp_bytes = tb;
        }
        tb = b->tmp_bytes;
        BWAMS_HIP(rocprim::radix_sort_pairs(b->d_tmp, tb, b->d_keys, b->d_keys2, b->d_vals, b->d_vals2, (size_t)n, 0,
                                            32 + rid_bits, st));
        launch_gather_sorted(b->d_pool, b->d_vals2, n, b->d_sorted, nullptr, opt->max_occ, st);
    }
    BWAMS_HIP(hipEventRecord(b->ev[4], st));
    BWAMS_HIP(hipEventRecord(b->ev[12], st));
    launch_ert_locate(e->t, b->d_enc, b->d_cum, b->d_sorted, n, with_sa ? b->d_sa_cnt : nullptr, opt->max_occ, b->d_ctr, b->d_ert_stk,
                      b->ert_stk_frames, b->cu_count, st);
    BWAMS_HIP(hipEventRecord(b->ev[13], st));
    if (with_sa && n > 0) {
        size_t tb = 0;
        BWAMS_HIP(rocprim::exclusive_scan(nullptr, tb, b->d_sa_cnt, b->d_sa_off, (int64_t)0, (size_t)n + 1,
                                          rocprim::plus<int64_t>(), st));
        if (tb > b->tmp_bytes) {
            BWAMS_HIP(hipStreamSynchronize(st));
            (void)hipFree(b->d_tmp);
            b->d_tmp = nullptr;
            BWAMS_HIP(dev_malloc(&b->d_tmp, tb));
            b->tmp_bytes = tb;
        }
        tb = b->tmp_bytes;
        BWAMS_HIP(hipMemsetAsync(b->d_sa_cnt + n, 0, 8, st));
        BWAMS_HIP(rocprim::exclusive_scan(b->d_tmp, tb, b->d_sa_cnt, b->d_sa_off, (int64_t)0, (size_t)n + 1,
                                          rocprim::plus<int64_t>(), st));
        if (int rrc = ert_redo_ensure(b, n)) return rrc;
        launch_ert_gather(e->t, b->d_sorted, n, b->d_sa_off, b->d_sa_coord, b->max_sa, opt->max_occ, b->d_ctr, b->d_ert_stk,
                          b->ert_stk_frames, b->d_ert_redo, b->max_sa, b->cu_count, st);
    }
    launch_ert_clear(b->d_sorted, n, st);
    BWAMS_HIP(hipEventRecord(b->ev[5], st));
    BWAMS_HIP(hipGetLastError());
    b->seed_done = true;
    return BWAMS_OK;
}

int bwams_seed_run_ert(bwams_batch_t *b, bwams_ert_t *e, const bwams_seed_opt_t *opt, int with_sa) {
    if (!b || !e || !opt) return BWAMS_ERR_ARG;
    if (e->idx != b->idx) {
        set_last_error("bwams_seed_run_ert: table and batch belong to different indexes");
        return BWAMS_ERR_ARG;
    }
    const int M = opt->split_width + 1 > opt->max_mem_intv ? opt->split_width + 1 : opt->max_mem_intv;
    if (opt->min_seed_len < e->t.K + e->t.X || M > 20 || M < 1) {
        set_last_error("bwams_seed_run_ert: needs min_seed_len >= kmer + xmer size, split_width < 20 and max_mem_intv <= 20 "
                       "(the trees store hit counts below 20 only)");
        return BWAMS_ERR_UNSUPPORTED;
    }
    if (b->max_read_len > 255 || b->max_read_len > e->t.read_len) {
        set_last_error("bwams_seed_run_ert: a read is longer than the read length the ERT was built for");
        return BWAMS_ERR_UNSUPPORTED;
    }
    b->last_seed_opt = *opt;
    b->seed_ert = e;
    int rc = ert_run_once(b, e, opt, with_sa, M);
    if (rc == BWAMS_ERR_CAPACITY && (b->n_smem > b->max_smem || b->n_pool_slots > b->pool_cap)) {
        BWAMS_HIP(hipStreamSynchronize(b->stream));
        // the slots handed out (holes included) bound what the chunk needs whatever filled the pool
        const int64_t seen = std::max(b->n_smem, b->n_pool_slots - seed_pool_slack(b->cu_count));
        const int64_t need = std::max(seen, b->max_smem) + seen / 4 + 1024;
        if ((rc = alloc_smem_buffers(b, need))) return rc;
        b->tmp_bytes = 0;
        if (b->d_tmp) { (void)hipFree(b->d_tmp); b->d_tmp = nullptr; }
        rc = ert_run_once(b, e, opt, with_sa, M);
    }
    return rc;
}

/* -------------------------------------------------------------- extension -- */

int bwams_bsw_upload(bwams_batch_t *b, const bwams_seqpair_t *pairs, int64_t n, const uint8_t *ref,
                     int64_t ref_bytes, const uint8_t *qer, int64_t qer_bytes) {
    if (!b || n < 0 || (n && (!pairs || !ref || !qer))) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    int qmax = 1, tmax = 1;
    for (int64_t i = 0; i < n; ++i) {
        const bwams_seqpair_t &p = pairs[i];
        if (p.len1 < 0 || p.len2 < 0 || p.idr < 0 || p.idq < 0 || (int64_t)p.idr + p.len1 > ref_bytes ||
            (int64_t)p.idq + p.len2 > qer_bytes) {
            set_last_error("bwams_bsw_upload: pair " + std::to_string(i) + " points outside the sequence buffers");
            return BWAMS_ERR_ARG;
        }
        if (p.len2 > qmax) qmax = p.len2;
        if (p.len1 > tmax) tmax = p.len1;
    }
    if (bsw_lds_bytes(qmax) > 160 * 1024) {
        set_last_error("bwams_bsw_upload: query longer than the LDS-resident kernel supports");
        return BWAMS_ERR_UNSUPPORTED;
    }
    auto grow = [](void **p, int64_t *cap, int64_t need, size_t elem) -> hipError_t {
        if (need <= *cap) return hipSuccess;
        if (*p) (void)hipFree(*p);
        *p = nullptr;
        *cap = need + need / 4 + 1024;
        return dev_malloc(p, (size_t)*cap * elem);
    };
    BWAMS_HIP(grow((void **)&b->d_pairs, &b->cap_pairs, n, sizeof(bwams_seqpair_t)));
    BWAMS_HIP(grow((void **)&b->d_ref, &b->cap_ref, ref_bytes + 64, 1));
    BWAMS_HIP(grow((void **)&b->d_qer, &b->cap_qer, qer_bytes + 64, 1));
    if (n) {
        BWAMS_HIP(hipMemcpyAsync(b->d_pairs, pairs, (size_t)n * sizeof(bwams_seqpair_t), hipMemcpyHostToDevice, b->stream));
        BWAMS_HIP(hipMemcpyAsync(b->d_ref, ref, (size_t)ref_bytes, hipMemcpyHostToDevice, b->stream));
        BWAMS_HIP(hipMemcpyAsync(b->d_qer, qer, (size_t)qer_bytes, hipMemcpyHostToDevice, b->stream));
        BWAMS_HIP(hipStreamSynchronize(b->stream));
    }
    b->n_pairs = n;
    b->max_qlen = qmax;
    b->max_tlen = tmax;
    return BWAMS_OK;
}

int bwams_bsw_run(bwams_batch_t *b, int32_t w, const bwams_sw_opt_t *o) {
    if (!b || !o) return BWAMS_ERR_ARG;
    if (o->e_ins <= 0 || o->e_del <= 0) {
        set_last_error("bwams_bsw_run: gap extension penalties must be positive");
        return BWAMS_ERR_ARG;
    }
    BWAMS_HIP(hipSetDevice(b->idx->device));
    SwParams prm;
    prm.o_del = o->o_del; prm.e_del = o->e_del; prm.o_ins = o->o_ins; prm.e_ins = o->e_ins;
    prm.zdrop = o->zdrop; prm.end_bonus = o->end_bonus;
    int mx = 0;
    for (int i = 0; i < 25; ++i) {
        prm.mat[i] = o->mat[i];
        mx = mx > o->mat[i] ? mx : o->mat[i];
    }
    prm.max_sc = mx;
    BWAMS_HIP(hipMemsetAsync(&b->d_ctr->bsw_cells, 0, sizeof(unsigned long long), b->stream));
    if (int lrc = bsw_list_ensure(b, b->n_pairs)) return lrc;
    BWAMS_HIP(hipEventRecord(b->ev[6], b->stream));
    if (launch_bsw(b->d_pairs, b->n_pairs, b->d_ref, b->d_qer, w, prm, b->max_qlen, b->d_ctr, b->cu_count, b->stream, b->d_bsw_list)) {
        set_last_error("bwams_bsw_run: a query longer than ~18000 bases does not fit the LDS kernel");
        return BWAMS_ERR_UNSUPPORTED;
    }
    BWAMS_HIP(hipEventRecord(b->ev[7], b->stream));
    BWAMS_HIP(hipGetLastError());
    return BWAMS_OK;
}

int bwams_bsw_fetch(bwams_batch_t *b, bwams_seqpair_t *pairs, int64_t n) {
    if (!b || n != b->n_pairs) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    if (n) BWAMS_HIP(hipMemcpyAsync(pairs, b->d_pairs, (size_t)n * sizeof(bwams_seqpair_t), hipMemcpyDeviceToHost, b->stream));
    BWAMS_HIP(hipStreamSynchronize(b->stream));
    return BWAMS_OK;
}

int bwams_bsw_extend(bwams_batch_t *b, bwams_seqpair_t *pairs, int64_t n, const uint8_t *ref, int64_t ref_bytes,
                     const uint8_t *qer, int64_t qer_bytes, int32_t w, const bwams_sw_opt_t *opt) {
    int rc = bwams_bsw_upload(b, pairs, n, ref, ref_bytes, qer, qer_bytes);
    if (rc) return rc;
    rc = bwams_bsw_run(b, w, opt);
    if (rc) return rc;
    return bwams_bsw_fetch(b, pairs, n);
}

/* -------------------------------------------------------------------- EMF ---- */

int bwams_emf_from_host(bwams_index_t *ix, int32_t seed_len, uint32_t seq_len, const uint32_t *loc_table,
                        uint32_t num_loc_entry, const bwams_seed_entry_t *seed_table, uint32_t num_seed_entry,
                        bwams_emf_t **out) {
    if (!ix || !out || !seed_table || !num_seed_entry || seed_len <= 0 || (num_loc_entry && !loc_table)) return BWAMS_ERR_ARG;
    if (!ix->d_ref) {
        set_last_error("bwams_emf_from_host: the index was opened without its .0123 reference");
        return BWAMS_ERR_ARG;
    }
    BWAMS_HIP(hipSetDevice(ix->device));
    bwams_emf *e = new bwams_emf();
    e->idx = ix;
    const size_t bs = (size_t)num_seed_entry * 16, bl = (size_t)(num_loc_entry ? num_loc_entry : 1) * 4;
    hipError_t he = dev_malloc(&e->d_seeds, bs);
    if (he == hipSuccess) he = dev_malloc(&e->d_loc, bl);
    if (he == hipSuccess) he = hipMemcpy(e->d_seeds, seed_table, bs, hipMemcpyHostToDevice);
    if (he == hipSuccess && num_loc_entry) he = hipMemcpy(e->d_loc, loc_table, (size_t)num_loc_entry * 4, hipMemcpyHostToDevice);
    if (he != hipSuccess) {                     // a table is tens of GiB: do not strand the half that was made
        set_last_error(std::string("bwams_emf_from_host: ") + hipGetErrorString(he));
        bwams_emf_close(e);
        return he == hipErrorOutOfMemory ? BWAMS_ERR_NOMEM : BWAMS_ERR_DEVICE;
    }
    e->t.seed_table = reinterpret_cast<const uint4 *>(e->d_seeds);
    e->t.loc_table = reinterpret_cast<const uint32_t *>(e->d_loc);
    e->t.ref = ix->fmi.ref;
    e->t.num_seed_entry = num_seed_entry;
    e->t.num_loc_entry = num_loc_entry;
    e->t.seq_len = seq_len;
    e->t.seed_len = seed_len;
    e->bytes = (int64_t)(bs + bl);
    *out = e;
    return BWAMS_OK;
}

int bwams_emf_open(bwams_index_t *ix, const char *path, bwams_emf_t **out) {
    if (!ix || !path || !out) return BWAMS_ERR_ARG;
    int fd = open(path, O_RDONLY);
    if (fd < 0) {
        set_last_error(std::string("cannot open ") + path);
        return BWAMS_ERR_IO;
    }
    struct stat st;
    fstat(fd, &st);
    const size_t fsz = (size_t)st.st_size;
    const uint8_t *m = fsz >= 64 ? (const uint8_t *)mmap(nullptr, fsz, PROT_READ, MAP_PRIVATE, fd, 0) : (const uint8_t *)MAP_FAILED;
    close(fd);
    if (m == MAP_FAILED) {
        set_last_error(std::string(path) + ": cannot map");
        return BWAMS_ERR_IO;
    }
    // perfect_table_t header (src/perfect.h:188-213)
    int32_t seed_len; uint32_t n_loc, n_seed, seq_len;
    memcpy(&seed_len, m, 4); memcpy(&n_loc, m + 4, 4); memcpy(&n_seed, m + 8, 4); memcpy(&seq_len, m + 40, 4);
    int rc;
    if (fsz != 64 + (size_t)n_loc * 4 + (size_t)n_seed * 16) {
        set_last_error(std::string(path) + ": size does not match its header");
        rc = BWAMS_ERR_IO;
    } else {
        rc = bwams_emf_from_host(ix, seed_len, seq_len, reinterpret_cast<const uint32_t *>(m + 64), n_loc,
                                 reinterpret_cast<const bwams_seed_entry_t *>(m + 64 + (size_t)n_loc * 4), n_seed, out);
    }
    munmap((void *)m, fsz);
    return rc;
}

int bwams_emf_from_device(bwams_index_t *ix, int32_t seed_len, uint32_t seq_len, const uint32_t *loc_table_dev,
                          uint32_t num_loc_entry, const bwams_seed_entry_t *seed_table_dev, uint32_t num_seed_entry,
                          bwams_emf_t **out) {
    if (!ix || !out || !seed_table_dev || !num_seed_entry || seed_len <= 0 || !ix->d_ref) return BWAMS_ERR_ARG;
    bwams_emf *e = new bwams_emf();
    e->idx = ix;
    e->owns = false;
    e->t.seed_table = reinterpret_cast<const uint4 *>(seed_table_dev);
    e->t.loc_table = loc_table_dev;
    e->t.ref = ix->fmi.ref;
    e->t.num_seed_entry = num_seed_entry;
    e->t.num_loc_entry = num_loc_entry;
    e->t.seq_len = seq_len;
    e->t.seed_len = seed_len;
    e->bytes = (int64_t)num_seed_entry * 16 + (int64_t)num_loc_entry * 4;
    *out = e;
    return BWAMS_OK;
}

/* Resident form: probe the reads uploaded by bwams_seed_upload and set the batch's skip flags on the
 * device, so that the following bwams_seed_run leaves the matched reads out. */
int bwams_emf_run(bwams_batch_t *b, bwams_emf_t *e) {
    if (!b || !e || e->idx != b->idx) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    const int64_t nseq = b->nseq;
    if (nseq > b->cap_emf) {
        if (b->d_emf_out) (void)hipFree(b->d_emf_out);
        if (b->d_emf_code) (void)hipFree(b->d_emf_code);
        b->d_emf_out = nullptr; b->d_emf_code = nullptr;
        b->cap_emf = nseq + nseq / 8 + 256;
        BWAMS_HIP(dev_malloc(&b->d_emf_out, (size_t)b->cap_emf * 8));
        BWAMS_HIP(dev_malloc(&b->d_emf_code, (size_t)b->cap_emf));
    }
    hipStream_t st = b->stream;
    BWAMS_HIP(hipMemsetAsync(&b->d_ctr->emf_nodes, 0, 16, st));
    BWAMS_HIP(hipEventRecord(b->ev_emf[0], st));
    launch_emf_probe(e->t, b->d_enc, b->d_cum, nseq, b->d_emf_out, b->d_emf_code, b->d_skip, b->d_ctr, st);
    BWAMS_HIP(hipEventRecord(b->ev_emf[1], st));
    // seed_run clears the counters: keep the probe's own
    BWAMS_HIP(hipMemcpyAsync(&b->h_ctr->emf_nodes, &b->d_ctr->emf_nodes, 16, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    b->emf_nodes = b->h_ctr->emf_nodes;
    b->emf_cmp_bytes = b->h_ctr->emf_cmp_bytes;
    BWAMS_HIP(hipGetLastError());
    b->has_skip = true;
    return BWAMS_OK;
}

int bwams_emf_fetch(bwams_batch_t *b, bwams_perfect_t *out, uint8_t *code) {
    if (!b || !b->d_emf_out) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    if (b->nseq) {
        if (out) BWAMS_HIP(hipMemcpyAsync(out, b->d_emf_out, (size_t)b->nseq * 8, hipMemcpyDeviceToHost, b->stream));
        if (code) BWAMS_HIP(hipMemcpyAsync(code, b->d_emf_code, (size_t)b->nseq, hipMemcpyDeviceToHost, b->stream));
    }
    BWAMS_HIP(hipStreamSynchronize(b->stream));
    return BWAMS_OK;
}

int bwams_emf_build(bwams_index_t *ix, int32_t seed_len, double slack, bwams_emf_t **out) {
    if (!ix || !out || seed_len < 16 || seed_len > 255 || !(slack >= 1.0 && slack <= 4.0)) {
        set_last_error("bwams_emf_build: seed length must be in [16, 255], slack in [1, 4]");
        return BWAMS_ERR_ARG;
    }
    if (!ix->d_ref) {
        set_last_error("bwams_emf_build: the index holds no .0123 reference");
        return BWAMS_ERR_ARG;
    }
    BWAMS_HIP(hipSetDevice(ix->device));
    hipDeviceProp_t prop;
    BWAMS_HIP(hipGetDeviceProperties(&prop, ix->device));
    const int64_t l_pac = (ix->fmi.ref_seq_len - 1) / 2;
    if (l_pac < seed_len) {
        set_last_error("bwams_emf_build: the reference is shorter than the seed length");
        return BWAMS_ERR_ARG;
    }
    bwams_emf *e = new bwams_emf();
    e->idx = ix;
    int64_t st[4] = {0, 0, 0, 0};
    const int rc = emf_build_device(e, (const uint8_t *)ix->d_ref, l_pac, seed_len, slack, prop.multiProcessorCount, knobs().verbose != 0, st);
    if (rc) { bwams_emf_close(e); return rc; }
    e->n_used = st[0]; e->n_key = st[1]; e->n_other = st[2]; e->build_ms = st[3];
    *out = e;
    return BWAMS_OK;
}

int bwams_emf_info(const bwams_emf_t *e, int32_t *seed_len, uint32_t *num_seed_entry, uint32_t *num_loc_entry, int64_t *n_used, int64_t *n_key,
                   int64_t *build_ms) {
    if (!e) return BWAMS_ERR_ARG;
    if (seed_len) *seed_len = e->t.seed_len;
    if (num_seed_entry) *num_seed_entry = e->t.num_seed_entry;
    if (num_loc_entry) *num_loc_entry = e->t.num_loc_entry;
    if (n_used) *n_used = e->n_used;
    if (n_key) *n_key = e->n_key;
    if (build_ms) *build_ms = e->build_ms;
    return BWAMS_OK;
}

int bwams_emf_table_fetch(bwams_emf_t *e, uint32_t *loc_table, bwams_seed_entry_t *seed_table) {
    if (!e) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(e->idx->device));
    if (loc_table && e->t.num_loc_entry) BWAMS_HIP(hipMemcpy(loc_table, e->t.loc_table, (size_t)e->t.num_loc_entry * 4, hipMemcpyDeviceToHost));
    if (seed_table) BWAMS_HIP(hipMemcpy(seed_table, e->t.seed_table, (size_t)e->t.num_seed_entry * 16, hipMemcpyDeviceToHost));
    return BWAMS_OK;
}

/* <path> in the reference's `.perfect.<L>` layout (perfect.h:188-213): 64-byte header, loc_table, seed_table */
int bwams_emf_save(bwams_emf_t *e, const char *path) {
    if (!e || !path) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(e->idx->device));
    FILE *f = fopen(path, "wb");
    if (!f) { set_last_error(std::string("bwams_emf_save: cannot create ") + path); return BWAMS_ERR_IO; }
    unsigned char hdr[64];
    memset(hdr, 0, sizeof hdr);
    const int32_t sl = e->t.seed_len;
    const uint32_t a[3] = {e->t.num_loc_entry, e->t.num_seed_entry, e->t.num_seed_entry};
    const uint32_t b3[3] = {e->t.seq_len, (uint32_t)e->n_used, (uint32_t)e->n_key};
    memcpy(hdr, &sl, 4); memcpy(hdr + 4, a, 12); memcpy(hdr + 40, b3, 12);
    int rc = fwrite(hdr, 1, 64, f) == 64 ? BWAMS_OK : BWAMS_ERR_IO;
    const size_t chunk = (size_t)256 << 20;
    void *stage = nullptr;
    if (rc == BWAMS_OK && hipHostMalloc(&stage, chunk) != hipSuccess) rc = BWAMS_ERR_NOMEM;
    auto stream_out = [&](const void *src, size_t total) {
        size_t done = 0;
        while (rc == BWAMS_OK && done < total) {
            const size_t n = total - done < chunk ? total - done : chunk;
            if (hipMemcpy(stage, (const uint8_t *)src + done, n, hipMemcpyDeviceToHost) != hipSuccess) { rc = BWAMS_ERR_DEVICE; break; }
            if (fwrite(stage, 1, n, f) != n) { rc = BWAMS_ERR_IO; break; }
            done += n;
        }
    };
    if (rc == BWAMS_OK) stream_out(e->t.loc_table, (size_t)e->t.num_loc_entry * 4);
    if (rc == BWAMS_OK) stream_out(e->t.seed_table, (size_t)e->t.num_seed_entry * 16);
    if (stage) (void)hipHostFree(stage);
    fclose(f);
    if (rc) set_last_error(std::string("bwams_emf_save: writing ") + path + " failed");
    return rc;
}

int bwams_emf_close(bwams_emf_t *e) {
    if (!e) return BWAMS_OK;
    if (!e->owns) { delete e; return BWAMS_OK; }
    (void)hipSetDevice(e->idx->device);
    if (e->d_seeds) (void)hipFree(e->d_seeds);
    if (e->d_loc) (void)hipFree(e->d_loc);
    delete e;
    return BWAMS_OK;
}

int bwams_emf_probe(bwams_batch_t *b, bwams_emf_t *e, const uint8_t *enc, const int64_t *cum, int64_t nseq,
                    bwams_perfect_t *out, uint8_t *code) {
    if (!b || !e || !cum || nseq < 0 || (nseq && (!enc || !out || !code))) return BWAMS_ERR_ARG;
    if (e->idx != b->idx) {
        set_last_error("bwams_emf_probe: table and batch belong to different indexes");
        return BWAMS_ERR_ARG;
    }
    const int64_t nb = cum[nseq] - cum[0];
    if (cum[0] != 0 || nseq > b->max_reads || nb > b->max_bases) return BWAMS_ERR_CAPACITY;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    if (nseq > b->cap_emf) {
        if (b->d_emf_out) (void)hipFree(b->d_emf_out);
        if (b->d_emf_code) (void)hipFree(b->d_emf_code);
        b->d_emf_out = nullptr; b->d_emf_code = nullptr;
        b->cap_emf = nseq + nseq / 8 + 256;
        BWAMS_HIP(dev_malloc(&b->d_emf_out, (size_t)b->cap_emf * 8));
        BWAMS_HIP(dev_malloc(&b->d_emf_code, (size_t)b->cap_emf));
    }
    hipStream_t st = b->stream;
    if (nb) BWAMS_HIP(hipMemcpyAsync(b->d_enc, enc, (size_t)nb, hipMemcpyHostToDevice, st));
    BWAMS_HIP(hipMemcpyAsync(b->d_cum, cum, (size_t)(nseq + 1) * 8, hipMemcpyHostToDevice, st));
    launch_emf_probe(e->t, b->d_enc, b->d_cum, nseq, b->d_emf_out, b->d_emf_code, nullptr, nullptr, st);
    BWAMS_HIP(hipGetLastError());
    if (nseq) {
        BWAMS_HIP(hipMemcpyAsync(out, b->d_emf_out, (size_t)nseq * 8, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipMemcpyAsync(code, b->d_emf_code, (size_t)nseq, hipMemcpyDeviceToHost, st));
    }
    BWAMS_HIP(hipStreamSynchronize(st));
    b->seed_done = false;            // the resident reads were replaced
    return BWAMS_OK;
}

/* ------------------------------------------------------------ mate rescue ---- */

int bwams_ksw_align(bwams_batch_t *b, const bwams_seqpair_t *pairs, int64_t n, const uint8_t *ref, int64_t ref_bytes,
                    const uint8_t *qer, int64_t qer_bytes, const bwams_sw_opt_t *o, bwams_kswr_t *out) {
    if (!b || !o || (n && !out)) return BWAMS_ERR_ARG;
    int mx = -128, mn = 127;
    for (int i = 0; i < 25; ++i) {
        mx = mx > o->mat[i] ? mx : o->mat[i];
        mn = mn < o->mat[i] ? mn : o->mat[i];
    }
    if (mx <= 0 || o->e_ins <= 0 || o->e_del <= 0 ||
        (o->o_ins + o->e_ins) + (o->o_del + o->e_del) <= mx - mn) {
        set_last_error("bwams_ksw_align: needs max(mat) > 0 and oe_ins + oe_del > max(mat) - min(mat) "
                       "(an insertion directly followed by a deletion must not beat a mismatch)");
        return BWAMS_ERR_UNSUPPORTED;
    }
    int rc = bwams_bsw_upload(b, pairs, n, ref, ref_bytes, qer, qer_bytes);
    if (rc) return rc;
    if (b->max_qlen > 512 || b->max_tlen > kKswMaxTarget) {
        set_last_error("bwams_ksw_align: query longer than 512 or target longer than 20000 "
                       "(the reference's kswv bounds are 512 / 2048, src/kswv.h:54-55)");
        return BWAMS_ERR_UNSUPPORTED;
    }
    BWAMS_HIP(hipSetDevice(b->idx->device));
    if (n > b->cap_ksw) {
        if (b->d_ksw_out) (void)hipFree(b->d_ksw_out);
        b->d_ksw_out = nullptr;
        b->cap_ksw = n + n / 4 + 256;
        BWAMS_HIP(dev_malloc(&b->d_ksw_out, (size_t)b->cap_ksw * sizeof(bwams_kswr_t)));
    }
    SwParams prm;
    prm.o_del = o->o_del; prm.e_del = o->e_del; prm.o_ins = o->o_ins; prm.e_ins = o->e_ins;
    prm.zdrop = o->zdrop; prm.end_bonus = o->end_bonus; prm.max_sc = mx;
    for (int i = 0; i < 25; ++i) prm.mat[i] = o->mat[i];
    BWAMS_HIP(hipEventRecord(b->ev[14], b->stream));
    if (launch_ksw(b->d_pairs, n, b->d_ref, b->d_qer, prm, ((b->max_qlen + 15) / 16) * 16, b->max_tlen, b->d_ksw_out,
                   b->d_ctr, b->cu_count, b->stream)) {
        set_last_error("bwams_ksw_align: target too long for the LDS of one block");
        return BWAMS_ERR_UNSUPPORTED;
    }
    BWAMS_HIP(hipEventRecord(b->ev[15], b->stream));
    BWAMS_HIP(hipGetLastError());
    if (n) BWAMS_HIP(hipMemcpyAsync(out, b->d_ksw_out, (size_t)n * sizeof(bwams_kswr_t), hipMemcpyDeviceToHost, b->stream));
    BWAMS_HIP(hipStreamSynchronize(b->stream));
    return BWAMS_OK;
}

/* ---------------------------------------------------------------- stats ---- */

int bwams_batch_stats(bwams_batch_t *b, bwams_stats_t *out) {
    if (!b || !out) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    BWAMS_HIP(hipMemcpyAsync(b->h_ctr, b->d_ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, b->stream));
    BWAMS_HIP(hipStreamSynchronize(b->stream));
    bwams_stats_t s;
    memset(&s, 0, sizeof s);
    const DevCounters &c = *b->h_ctr;
    s.n_ext = (int64_t)c.n_ext;
    s.n_ext_blocks = (int64_t)c.n_ext_blocks;
    s.n_sa_lookups = (int64_t)c.n_sa_lookups;
    s.n_lf_steps = (int64_t)c.n_lf_steps;
    s.n_smem[0] = (int64_t)c.valid_after[0];
    s.n_smem[1] = (int64_t)(c.valid_after[1] - c.valid_after[0]);
    s.n_smem[2] = (int64_t)(c.valid_after[2] - c.valid_after[1]);
    s.bsw_cells = (int64_t)c.bsw_cells;
    for (int i = 0; i < 3; ++i) {
        s.n_ext_round[i] = (int64_t)(c.ext_after[i] - (i ? c.ext_after[i - 1] : 0));
        s.n_blk_round[i] = (int64_t)(c.blk_after[i] - (i ? c.blk_after[i - 1] : 0));
    }
    auto el = [&](int a, int bb, float *dst) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, b->ev[a], b->ev[bb]) == hipSuccess) *dst = ms;
    };
    if (b->seed_done) {
        el(8, 9, &s.ms_smem_r1);
        el(10, 11, &s.ms_smem_r2);
        el(12, 13, &s.ms_smem_r3);
        el(3, 4, &s.ms_sort);
        el(4, 5, &s.ms_sal);
        el(0, 5, &s.ms_seed_total);
    }
    el(6, 7, &s.ms_bsw);
    el(14, 15, &s.ms_ksw);
    el(1, 2, &s.ms_tasks);
    {
        float ms = 0;
        if (hipEventElapsedTime(&ms, b->ev_emf[0], b->ev_emf[1]) == hipSuccess) s.ms_emf = ms;
    }
    s.ert_kmer_lookups = (int64_t)c.ert_kmer;
    s.ert_node_reads = (int64_t)c.ert_nodes;
    s.ert_ref_bytes = (int64_t)c.ert_ref;
    s.emf_nodes = (int64_t)b->emf_nodes;
    s.emf_cmp_bytes = (int64_t)b->emf_cmp_bytes;
    chain_state_stats(b->chain, &s);
    (void)hipGetLastError();
    *out = s;
    return BWAMS_OK;
}

}  // extern "C"
