"""ctypes binding of the C-ABI in include/bwams.h (libbwams.so).

Used by the tests and bench.py; a reference-side caller would bind the same
symbols from C++.  There is no fallback: if the library or a
gfx950 device is missing, calls raise BwamsError.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("BWAMS_LIB") or os.path.join(PKG, "_build", "libbwams.so")     # BWAMS_LIB: A/B runs of another build

SMEM_DTYPE = np.dtype([("rid", "<u4"), ("m", "<u4"), ("n", "<u4"), ("pad_", "<u4"),
                       ("k", "<i8"), ("l", "<i8"), ("s", "<i8")])
SEQPAIR_DTYPE = np.dtype([(n, "<i4") for n in
                          ("idr", "idq", "id", "len1", "len2", "h0", "seqid", "regid",
                           "score", "tle", "gtle", "qle", "gscore", "max_off")])

# every symbol include/bwams.h declares
SYMBOLS = [
    "bwams_strerror", "bwams_last_error", "bwams_device_count", "bwams_debug_reload",
    "bwams_index_open", "bwams_index_from_host", "bwams_index_from_device", "bwams_index_close",
    "bwams_index_bytes", "bwams_index_build", "bwams_index_fetch", "bwams_index_save", "bwams_reg2aln_run", "bwams_reg2aln_run_sam", "bwams_reg2aln_fetch",
    "bwams_index_set_contig_names", "bwams_index_set_contig_annos", "bwams_sam_upload", "bwams_sam_run", "bwams_sam_run_pe", "bwams_sam_run_emf", "bwams_sam_fetch",
    "bwams_process_chunk", "bwams_process_chunk_smart", "bwams_process_chunk2", "bwams_emf_regs_merge", "bwams_fastq_decode", "bwams_fastq_info", "bwams_fastq_has_qual", "bwams_fastq_fetch", "bwams_fastq_to_batch", "bwams_fastq_to_batch_opt", "bwams_fastq_close", "bwams_batch_create", "bwams_batch_destroy", "bwams_seed_fmi",
    "bwams_seed_upload", "bwams_seed_run", "bwams_seed_counts", "bwams_seed_fetch",
    "bwams_ert_from_host", "bwams_ert_open", "bwams_ert_close", "bwams_ert_bytes", "bwams_ert_set_fat", "bwams_seed_run_ert",
    "bwams_ert_build", "bwams_ert_info", "bwams_ert_fetch", "bwams_ert_save", "bwams_debug_sort",
    "bwams_emf_build", "bwams_emf_info", "bwams_emf_table_fetch", "bwams_emf_save",
    "bwams_bsw_extend", "bwams_bsw_upload", "bwams_bsw_run", "bwams_bsw_fetch",
    "bwams_batch_stats", "bwams_batch_sync", "bwams_ksw_align",
    "bwams_index_build_fma", "bwams_index_set_fma", "bwams_index_fetch_fma",
    "bwams_emf_open", "bwams_emf_from_host", "bwams_emf_close", "bwams_emf_probe",
    "bwams_emf_from_device", "bwams_emf_run", "bwams_emf_fetch",
    "bwams_index_set_contigs", "bwams_chain_run", "bwams_chain_fetch", "bwams_chain_upload",
    "bwams_extend_build", "bwams_extend_run", "bwams_extend_fetch", "bwams_extend_tasks_fetch",
    "bwams_process_reads", "bwams_process_reads_stage1", "bwams_process_reads_stage2", "bwams_host_alloc", "bwams_host_free",
    "bwams_reader_open", "bwams_reader_next", "bwams_reader_release", "bwams_reader_error", "bwams_reader_close",
    "bwams_writer_open", "bwams_writer_put", "bwams_writer_close",
    "bwams_process_reads_upload", "bwams_process_reads_stage1_run", "bwams_batch_device", "bwams_multi_upload", "bwams_multi_compute",
    "bwams_shard_bounds", "bwams_multi_create", "bwams_multi_process_reads", "bwams_multi_fetch", "bwams_multi_error", "bwams_multi_destroy",
    "bwams_dedup_run", "bwams_dedup_fetch", "bwams_chain_run_ert", "bwams_pestat", "bwams_pestat_keys", "bwams_pestat_from_keys", "bwams_pair_run", "bwams_pair_run_sam", "bwams_pair_fetch", "bwams_emf_regs_run", "bwams_emf_regs_fetch",
]
ERT_MEM_DTYPE = np.dtype([("forward", "u1"), ("pad_", "u1", (3,)), ("start", "<i4"), ("end", "<i4"), ("rc_start", "<i4"),
                          ("rc_end", "<i4"), ("skip_ref_fetch", "<i4"), ("fetch_leaves", "<i4"), ("hitbeg", "<i4"),
                          ("hitcount", "<i4"), ("end_correction", "<i4"), ("is_multi_hit", "<i4"), ("c_pivot", "<i4"),
                          ("p_pivot", "<i4"), ("pp_pivot", "<i4")])
assert ERT_MEM_DTYPE.itemsize == 56
PAIR_DTYPE = np.dtype([("score", "<i4"), ("sub", "<i4"), ("n_sub", "<i4"), ("z", "<i4", (2,)), ("n_pri", "<i4", (2,)),
                       ("n_matesw", "<i4")])
assert PAIR_DTYPE.itemsize == 32
PESTAT_DTYPE = np.dtype([("low", "<i4"), ("high", "<i4"), ("failed", "<i4"), ("pad_", "<i4"), ("avg", "<f8"), ("std", "<f8")])

# records of include/bwams_types.h (layouts of bntann1_t's subset, mem_seed_t, mem_chain_t, mem_alnreg_t)
CONTIG_DTYPE = np.dtype([("offset", "<i8"), ("len", "<i4"), ("is_alt", "<i4")])
CHAIN_SEED_DTYPE = np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4"), ("score", "<i4"), ("done", "i1"),
                             ("pad0_", "i1", 3), ("aln", "<i4"), ("pad1_", "<i4")])
CHAIN_DTYPE = np.dtype([("seqid", "<i4"), ("cseed", "<i4"), ("n", "<i4"), ("m", "<i4"), ("first", "<i4"), ("rid", "<i4"),
                        ("w_kept_alt", "<u4"), ("frac_rep", "<f4"), ("pos", "<i8"), ("seed_off", "<i8")])
ALNREG_DTYPE = np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("rid", "<i4"), ("pad0_", "<i4"),
                         ("chain", "<i8"), ("score", "<i4"), ("truesc", "<i4"), ("sub", "<i4"), ("alt_sc", "<i4"),
                         ("csub", "<i4"), ("sub_n", "<i4"), ("w", "<i4"), ("seedcov", "<i4"), ("secondary", "<i4"),
                         ("secondary_all", "<i4"), ("seedlen0", "<i4"), ("n_comp_is_alt", "<i4"), ("frac_rep", "<f4"),
                         ("pad1_", "<i4"), ("hash", "<u8"), ("flg", "<i4"), ("pad2_", "<i4")])
ALN_DTYPE = np.dtype([("pos", "<i8"), ("rid", "<i4"), ("flag", "<i4"), ("is_rev", "<i4"), ("is_alt", "<i4"), ("mapq", "<i4"),
                      ("NM", "<i4"), ("n_cigar", "<i4"), ("md_len", "<i4"), ("cigar_off", "<i8"), ("md_off", "<i8"),
                      ("score", "<i4"), ("sub", "<i4"), ("alt_sc", "<i4"), ("pad_", "<i4")])
assert ALN_DTYPE.itemsize == 72
assert CONTIG_DTYPE.itemsize == 16 and CHAIN_SEED_DTYPE.itemsize == 32 and CHAIN_DTYPE.itemsize == 48
assert ALNREG_DTYPE.itemsize == 112


class BwamsError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        super().__init__(f"{where}: error {code} {detail}")


class SeedOpt(C.Structure):
    _fields_ = [("min_seed_len", C.c_int32), ("split_factor", C.c_float),
                ("split_width", C.c_int32), ("max_mem_intv", C.c_int32), ("max_occ", C.c_int32)]


class SwOpt(C.Structure):
    _fields_ = [("o_del", C.c_int32), ("e_del", C.c_int32), ("o_ins", C.c_int32),
                ("e_ins", C.c_int32), ("zdrop", C.c_int32), ("end_bonus", C.c_int32),
                ("mat", C.c_int8 * 25), ("pad_", C.c_int8 * 3)]


class MemOpt(C.Structure):
    """bwams_mem_opt_t: the subset of mem_opt_t read by chaining and chain-to-alignment."""
    _fields_ = [("a", C.c_int32), ("o_del", C.c_int32), ("e_del", C.c_int32), ("o_ins", C.c_int32), ("e_ins", C.c_int32),
                ("pen_clip5", C.c_int32), ("pen_clip3", C.c_int32), ("w", C.c_int32), ("zdrop", C.c_int32),
                ("min_seed_len", C.c_int32), ("min_chain_weight", C.c_int32), ("max_chain_extend", C.c_int32),
                ("max_occ", C.c_int32), ("max_chain_gap", C.c_int32), ("mask_level", C.c_float),
                ("drop_ratio", C.c_float), ("mat", C.c_int8 * 25), ("pad_", C.c_int8 * 3), ("extend_all", C.c_int32), ("mask_level_redun", C.c_float), ("max_ins", C.c_int32),
                ("b", C.c_int32), ("pen_unpaired", C.c_int32), ("max_matesw", C.c_int32), ("mapq_coef_len", C.c_int32)]


class FmiDesc(C.Structure):
    _fields_ = [("ref_seq_len", C.c_int64), ("count", C.c_int64 * 5),
                ("cp_occ", C.c_void_p), ("sa_ms_byte", C.c_void_p), ("sa_ls_word", C.c_void_p),
                ("sentinel_index", C.c_int64), ("ref_0123", C.c_void_p)]


class Fastq:
    """A decoded FASTQ buffer resident on the GPU (bwams_fastq_t)."""

    def __init__(self, text, device: int = 0, n_bytes: int | None = None):
        """text: bytes (host) or an int device address with n_bytes."""
        self.h = C.c_void_p()
        n, nb = C.c_int64(0), C.c_int64(0)
        if isinstance(text, int):
            rc = lib().bwams_fastq_decode(device, C.c_void_p(text), C.c_int64(n_bytes), C.byref(self.h), C.byref(n), C.byref(nb))
        else:
            rc = lib().bwams_fastq_decode(device, text, C.c_int64(len(text)), C.byref(self.h), C.byref(n), C.byref(nb))
        _chk(rc, "bwams_fastq_decode")
        self.n_reads, self.n_bases = n.value, nb.value

    def info(self):
        n, nb, nn, nc, ms = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_float(0)
        _chk(lib().bwams_fastq_info(self.h, C.byref(n), C.byref(nb), C.byref(nn), C.byref(nc), C.byref(ms)), "bwams_fastq_info")
        return dict(n_reads=n.value, n_bases=nb.value, name_bytes=nn.value, comment_bytes=nc.value, ms=ms.value)

    def fetch(self):
        i = self.info()
        n1 = i["n_reads"] + 1
        enc = np.zeros(max(i["n_bases"], 1), np.uint8); qual = np.zeros(max(i["n_bases"], 1), np.uint8)
        names = np.zeros(max(i["name_bytes"], 1), np.uint8); comm = np.zeros(max(i["comment_bytes"], 1), np.uint8)
        cum = np.zeros(n1, np.int64); noff = np.zeros(n1, np.int64); coff = np.zeros(n1, np.int64)
        _chk(lib().bwams_fastq_fetch(self.h, _p(enc), _p(cum), _p(names), _p(noff), _p(qual), _p(comm), _p(coff)), "bwams_fastq_fetch")
        nb_, cb_ = bytes(names[:i["name_bytes"]]), bytes(comm[:i["comment_bytes"]])
        return dict(n=i["n_reads"], enc=enc[:i["n_bases"]], cum=cum, quals=qual[:i["n_bases"]] if lib().bwams_fastq_has_qual(self.h) else None,
                    names=[nb_[noff[k]:noff[k + 1]] for k in range(i["n_reads"])],
                    comments=[cb_[coff[k]:coff[k + 1]] or None for k in range(i["n_reads"])])

    def to_batch(self, batch: "Batch"):
        """bwams_seed_upload + bwams_sam_upload of the decoded chunk, device to device."""
        _chk(lib().bwams_fastq_to_batch(self.h, batch.h), "bwams_fastq_to_batch")
        batch._nseq = self.n_reads

    def close(self):
        if self.h:
            lib().bwams_fastq_close(self.h)
            self.h = C.c_void_p()


class SamOpt(C.Structure):
    """bwams_sam_opt_t"""
    _fields_ = [("T", C.c_int32), ("flag", C.c_int32), ("XA_drop_ratio", C.c_float), ("max_XA_hits", C.c_int32),
                ("max_XA_hits_alt", C.c_int32), ("rg_id", C.c_char * 256)]


def default_sam_opt(flag: int = 0, rg_id: bytes = b"") -> SamOpt:
    """mem_opt_init defaults (src/bwamem.cpp:135-171)."""
    return SamOpt(30, flag, 0.80, 5, 200, rg_id)


class Stats(C.Structure):
    _fields_ = [("n_ext", C.c_int64), ("n_ext_blocks", C.c_int64), ("n_sa_lookups", C.c_int64),
                ("n_lf_steps", C.c_int64), ("n_smem", C.c_int64 * 3), ("bsw_cells", C.c_int64),
                ("n_ext_round", C.c_int64 * 3), ("n_blk_round", C.c_int64 * 3),
                ("ms_smem_r1", C.c_float), ("ms_smem_r2", C.c_float), ("ms_smem_r3", C.c_float),
                ("ms_sort", C.c_float), ("ms_sal", C.c_float), ("ms_seed_total", C.c_float),
                ("ms_bsw", C.c_float), ("ms_ksw", C.c_float), ("ms_tasks", C.c_float), ("ms_emf", C.c_float),
                ("emf_nodes", C.c_int64), ("emf_cmp_bytes", C.c_int64),
                ("n_chains", C.c_int64), ("n_chain_seeds", C.c_int64), ("n_left", C.c_int64), ("n_right", C.c_int64),
                ("n_retry_left", C.c_int64), ("n_retry_right", C.c_int64),
                ("ms_chain", C.c_float), ("ms_ext_plan", C.c_float), ("ms_ext_left", C.c_float),
                ("ms_ext_right", C.c_float), ("ms_ext_purge", C.c_float), ("ms_ext_total", C.c_float),
                ("n_ext_rounds", C.c_int64), ("n_final_regs", C.c_int64), ("ms_dedup", C.c_float), ("ms_pair", C.c_float),
                ("n_pair_tasks", C.c_int64), ("n_pair_redone", C.c_int64), ("n_pair_regs", C.c_int64), ("n_chain_redo", C.c_int64),
                ("ert_kmer_lookups", C.c_int64), ("ert_node_reads", C.c_int64), ("ert_ref_bytes", C.c_int64)]


class BuildStats(C.Structure):
    _fields_ = [("rows", C.c_int64), ("chunks", C.c_int32), ("rounds", C.c_int32), ("unresolved_after_first", C.c_int64),
                ("ms_first_pass", C.c_float), ("ms_outputs", C.c_float)]


def pestat_from_keys(keys) -> np.ndarray:
    """mem_pestat's arithmetic over the insert-size keys of a whole chunk (host only; keys in any order)."""
    keys = np.ascontiguousarray(keys, np.uint64)
    pes = np.zeros(4, PESTAT_DTYPE)
    _chk(lib().bwams_pestat_from_keys(_p(keys), len(keys), _p(pes)), "bwams_pestat_from_keys")
    return pes


def default_seed_opt() -> SeedOpt:
    """mem_opt_init defaults (/root/reference/src/bwamem.cpp:135-171)."""
    return SeedOpt(19, 1.5, 10, 20, 500)


def default_sw_opt(end_bonus: int = 5, a: int = 1, b: int = 4) -> SwOpt:
    o = SwOpt(6, 1, 6, 1, 100, end_bonus)
    k = 0
    for i in range(4):
        for j in range(4):
            o.mat[k] = a if i == j else -b
            k += 1
        o.mat[k] = -1
        k += 1
    for j in range(5):
        o.mat[k] = -1
        k += 1
    return o


def default_mem_opt(a: int = 1, b: int = 4) -> MemOpt:
    """mem_opt_init defaults (/root/reference/src/bwamem.cpp:135-171)."""
    o = MemOpt(a, 6, 1, 6, 1, 5, 5, 100, 100, 19, 0, 1 << 30, 500, 10000, 0.5, 0.5)
    o.mask_level_redun = 0.95
    o.max_ins = 10000
    o.b, o.pen_unpaired, o.max_matesw = b, 17, 50
    o.mapq_coef_len = 50
    sw = default_sw_opt(5, a, b)
    for i in range(25):
        o.mat[i] = sw.mat[i]
    return o


def build(force: bool = False) -> str:
    """Compile libbwams.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(PKG, "csrc")
    cmd = ["make", "-s", "-C", csrc, "-j4"]
    if force:
        subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    """Load libbwams.so (must have been built; raises if it is missing)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BwamsError(-1, "libbwams.so", f"not built: {LIB_PATH} (run __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
        L.bwams_strerror.restype = C.c_char_p
        L.bwams_strerror.argtypes = [C.c_int]
        L.bwams_last_error.restype = C.c_char_p
        L.bwams_device_count.argtypes = [vp]
        L.bwams_index_open.argtypes = [C.c_char_p, C.c_int, vp]
        L.bwams_index_from_host.argtypes = [vp, C.c_int, vp]
        L.bwams_index_from_device.argtypes = [vp, C.c_int, vp]
        L.bwams_index_close.argtypes = [vp]
        L.bwams_index_build.argtypes = [vp, i64, C.c_int, C.c_int, C.c_int, i64, vp, vp]
        L.bwams_index_fetch.argtypes = [vp, vp, vp, vp, vp, vp]
        L.bwams_index_save.argtypes = [vp, C.c_char_p]
        L.bwams_reg2aln_run.argtypes = [vp, vp, i32, vp, vp, vp]
        L.bwams_reg2aln_fetch.argtypes = [vp, vp, i64, vp, i64, vp, i64]
        L.bwams_index_build_fma.argtypes = [vp, C.c_int, C.c_int]
        L.bwams_index_set_fma.argtypes = [vp, vp, C.c_int, vp, C.c_int]
        L.bwams_index_fetch_fma.argtypes = [vp, vp, vp]
        L.bwams_index_bytes.restype = i64
        L.bwams_index_bytes.argtypes = [vp]
        L.bwams_batch_create.argtypes = [vp, i64, i64, i64, i64, vp]
        L.bwams_batch_destroy.argtypes = [vp]
        L.bwams_seed_fmi.argtypes = [vp, vp, vp, vp, i64, vp, vp, i64, vp, vp, i64, vp, vp]
        L.bwams_seed_upload.argtypes = [vp, vp, vp, vp, i64]
        L.bwams_index_set_contigs.argtypes = [vp, vp, i32]
        L.bwams_chain_run.argtypes = [vp, vp, vp, vp]
        L.bwams_chain_fetch.argtypes = [vp, vp, i64, vp, i64, vp]
        L.bwams_chain_upload.argtypes = [vp, vp, i64, vp, i64, vp]
        L.bwams_extend_build.argtypes = [vp, vp, vp, vp]
        L.bwams_extend_run.argtypes = [vp, vp, vp]
        L.bwams_extend_fetch.argtypes = [vp, vp, i64, vp, vp]
        L.bwams_extend_tasks_fetch.argtypes = [vp, i32, vp, i64, vp, i64, vp, i64, vp, vp, vp]
        L.bwams_dedup_run.argtypes = [vp, vp, vp]
        L.bwams_dedup_fetch.argtypes = [vp, vp, i64, vp]
        L.bwams_pestat.argtypes = [vp, vp, vp]
        L.bwams_chain_run_ert.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
        L.bwams_pestat_keys.argtypes = [vp, vp, vp, C.c_int64, vp]
        L.bwams_pestat_from_keys.argtypes = [vp, C.c_int64, vp]
        L.bwams_pair_run.argtypes = [vp, vp, vp, C.c_int64, C.c_int32, vp, vp]
        L.bwams_pair_fetch.argtypes = [vp, vp, C.c_int64, vp, vp]
        L.bwams_emf_regs_run.argtypes = [vp, vp, vp, vp]
        L.bwams_emf_regs_fetch.argtypes = [vp, vp, i64, vp, vp]
        L.bwams_seed_run.argtypes = [vp, vp, C.c_int]
        L.bwams_seed_counts.argtypes = [vp, vp, vp]
        L.bwams_seed_fetch.argtypes = [vp, vp, i64, vp, i64, vp]
        L.bwams_bsw_extend.argtypes = [vp, vp, i64, vp, i64, vp, i64, i32, vp]
        L.bwams_bsw_upload.argtypes = [vp, vp, i64, vp, i64, vp, i64]
        L.bwams_bsw_run.argtypes = [vp, i32, vp]
        L.bwams_bsw_fetch.argtypes = [vp, vp, i64]
        L.bwams_ksw_align.argtypes = [vp, vp, i64, vp, i64, vp, i64, vp, vp]
        L.bwams_emf_open.argtypes = [vp, C.c_char_p, vp]
        L.bwams_emf_from_host.argtypes = [vp, i32, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, vp]
        L.bwams_emf_close.argtypes = [vp]
        L.bwams_emf_from_device.argtypes = [vp, i32, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, vp]
        L.bwams_emf_run.argtypes = [vp, vp]
        L.bwams_emf_fetch.argtypes = [vp, vp, vp]
        L.bwams_emf_probe.argtypes = [vp, vp, vp, vp, i64, vp, vp]
        L.bwams_ert_from_host.argtypes = [vp, vp, i32, i32, i32, vp, i64, vp]
        L.bwams_ert_open.argtypes = [vp, C.c_char_p, i32, vp]
        L.bwams_ert_close.argtypes = [vp]
        L.bwams_ert_bytes.restype = i64
        L.bwams_ert_bytes.argtypes = [vp]
        L.bwams_seed_run_ert.argtypes = [vp, vp, vp, C.c_int]
        L.bwams_ert_build.argtypes = [vp, i32, i32, i32, i32, vp]
        L.bwams_ert_info.argtypes = [vp, vp, vp, vp, vp, vp]
        L.bwams_ert_fetch.argtypes = [vp, vp, vp]
        L.bwams_ert_save.argtypes = [vp, C.c_char_p]
        L.bwams_debug_sort.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
        L.bwams_emf_build.argtypes = [vp, i32, C.c_double, vp]
        L.bwams_emf_info.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.bwams_emf_table_fetch.argtypes = [vp, vp, vp]
        L.bwams_emf_save.argtypes = [vp, C.c_char_p]
        L.bwams_batch_stats.argtypes = [vp, vp]
        L.bwams_batch_sync.argtypes = [vp]
        _lib = L
    return _lib


def _chk(rc: int, where: str):
    if rc != 0:
        L = lib()
        raise BwamsError(rc, where, f"({L.bwams_strerror(rc).decode()}) {L.bwams_last_error().decode()}")


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Index:
    """FM-index resident in one GPU's HBM."""

    def __init__(self, handle, keep=None):
        self.h = handle
        self._keep = keep

    @classmethod
    def open(cls, prefix: str, device: int = 0) -> "Index":
        h = C.c_void_p()
        _chk(lib().bwams_index_open(prefix.encode(), device, C.byref(h)), "bwams_index_open")
        return cls(h)

    @classmethod
    def from_host(cls, idx, device: int = 0) -> "Index":
        """idx: bwams.fmindex.FMIndex of numpy arrays."""
        cp = np.ascontiguousarray(idx.cp_occ)
        ms = np.ascontiguousarray(idx.sa_ms_byte)
        ls = np.ascontiguousarray(idx.sa_ls_word)
        ref = np.ascontiguousarray(idx.ref_0123) if idx.ref_0123 is not None else None
        d = FmiDesc(int(idx.ref_seq_len), (C.c_int64 * 5)(*[int(x) for x in idx.count]),
                    cp.ctypes.data, ms.ctypes.data, ls.ctypes.data, int(idx.sentinel_index),
                    ref.ctypes.data if ref is not None else None)
        h = C.c_void_p()
        _chk(lib().bwams_index_from_host(C.byref(d), device, C.byref(h)), "bwams_index_from_host")
        return cls(h)

    @classmethod
    def from_device(cls, idx, device: int = 0) -> "Index":
        """idx: FMIndex whose arrays are torch tensors on `device` (kept alive here)."""
        ref = idx.ref_0123
        d = FmiDesc(int(idx.ref_seq_len), (C.c_int64 * 5)(*[int(x) for x in idx.count]),
                    idx.cp_occ.data_ptr(), idx.sa_ms_byte.data_ptr(), idx.sa_ls_word.data_ptr(),
                    int(idx.sentinel_index), ref.data_ptr() if ref is not None else None)
        h = C.c_void_p()
        _chk(lib().bwams_index_from_device(C.byref(d), device, C.byref(h)), "bwams_index_from_device")
        return cls(h, keep=idx)

    @classmethod
    def build(cls, genome, device: int = 0, keep_ref: bool = True, chunk_rows: int = 0) -> "Index":
        """FM-index of fw || revcomp(fw) built on the GPU (bwams_index_build).  genome: numpy uint8 codes 0..3, or a torch
        uint8 tensor already on `device`."""
        h = C.c_void_p()
        st = BuildStats()
        if isinstance(genome, np.ndarray):
            g = np.ascontiguousarray(genome, dtype=np.uint8)
            _chk(lib().bwams_index_build(_p(g), len(g), 0, device, int(keep_ref), chunk_rows, C.byref(st), C.byref(h)), "bwams_index_build")
        else:
            g = genome.contiguous()
            _chk(lib().bwams_index_build(g.data_ptr(), g.numel(), 1, device, int(keep_ref), chunk_rows, C.byref(st), C.byref(h)),
                 "bwams_index_build")
        ix = cls(h)
        ix.build_stats = st
        return ix

    def fetch(self, with_ref: bool = True):
        """The resident arrays as a bwams.fmindex.FMIndex of numpy arrays (for the file writer / the CPU oracle)."""
        from bwams import fmindex
        d = FmiDesc()
        _chk(lib().bwams_index_fetch(self.h, None, None, None, None, C.byref(d)), "bwams_index_fetch")
        L = int(d.ref_seq_len)
        cp = np.empty(((L >> 6) + 1, 8), dtype=np.uint64)
        ms = np.empty((L >> 3) + 1, dtype=np.int8)
        ls = np.empty((L >> 3) + 1, dtype=np.uint32)
        ref = np.empty(L - 1, dtype=np.uint8) if with_ref else None
        _chk(lib().bwams_index_fetch(self.h, _p(cp), _p(ms), _p(ls), _p(ref), C.byref(d)), "bwams_index_fetch")
        return fmindex.FMIndex(L, np.array(list(d.count), dtype=np.int64), cp, ms, ls, int(d.sentinel_index), ref)

    def save(self, prefix: str):
        _chk(lib().bwams_index_save(self.h, prefix.encode()), "bwams_index_save")

    def build_fma(self, all_bp: int = 11, last_bp: int = 13):
        _chk(lib().bwams_index_build_fma(self.h, all_bp, last_bp), "bwams_index_build_fma")
        self._fma = (all_bp, last_bp)

    def set_fma(self, all_tab, all_bp, last_tab, last_bp):
        if all_tab is None:
            _chk(lib().bwams_index_set_fma(self.h, None, 0, None, 0), "bwams_index_set_fma")
            return
        a = np.ascontiguousarray(all_tab); l = np.ascontiguousarray(last_tab)
        _chk(lib().bwams_index_set_fma(self.h, _p(a), all_bp, _p(l), last_bp), "bwams_index_set_fma")
        self._fma = (all_bp, last_bp)

    def fetch_fma(self):
        all_bp, last_bp = self._fma
        a = np.zeros((4 ** all_bp, 32), dtype=np.uint32)
        l = np.zeros((4 ** last_bp, 4), dtype=np.uint32)
        _chk(lib().bwams_index_fetch_fma(self.h, _p(a), _p(l)), "bwams_index_fetch_fma")
        return a, l

    def set_contigs(self, contigs):
        c = np.ascontiguousarray(contigs, dtype=CONTIG_DTYPE)
        _chk(lib().bwams_index_set_contigs(self.h, _p(c), len(c)), "bwams_index_set_contigs")

    def set_contig_names(self, names):
        """Sequence names for the SAM text (list of bytes / str, one per sequence set with set_contigs)."""
        blob, off = bytearray(), []
        for nm in names:
            off.append(len(blob))
            blob += (nm if isinstance(nm, bytes) else nm.encode()) + b"\0"
        off.append(len(blob))
        off = np.asarray(off, np.int32)
        _chk(lib().bwams_index_set_contig_names(self.h, bytes(blob), _p(off)), "bwams_index_set_contig_names")

    def set_contig_annos(self, annos):
        """bntann1_t.anno of every sequence (b"" = none): the XR:Z: tags of MEM_F_REF_HDR (0x100)."""
        blob, off = bytearray(), []
        for a in annos:
            off.append(len(blob))
            blob += (a if isinstance(a, bytes) else a.encode()) + b"\0"
        off.append(len(blob))
        off = np.asarray(off, np.int32)
        _chk(lib().bwams_index_set_contig_annos(self.h, bytes(blob), _p(off)), "bwams_index_set_contig_annos")

    def debug_sort(self, k, s, q, which: int, mode: int = 0):
        """order of the wave tier's region sort (test hook)"""
        k = np.ascontiguousarray(k, np.int64); s = np.ascontiguousarray(s, np.int32); q = np.ascontiguousarray(q, np.int32)
        out = np.zeros(len(k), np.int32)
        _chk(lib().bwams_debug_sort(self.h, _p(k), _p(s), _p(q), len(k), which, mode, _p(out)), "bwams_debug_sort")
        return out

    @property
    def nbytes(self) -> int:
        return lib().bwams_index_bytes(self.h)

    def close(self):
        if self.h:
            lib().bwams_index_close(self.h)
            self.h = None


class Emf:
    """EMF table resident in HBM (bwams.emf.EmfTable or a <prefix>.perfect.<L> file)."""

    def __init__(self, index: Index, table=None, path: str | None = None, device_table=None):
        self.index = index
        self.h = C.c_void_p()
        self._keep = device_table
        if device_table is not None:          # EmfTable of torch tensors on the index's GPU
            t = device_table
            _chk(lib().bwams_emf_from_device(index.h, t.seed_len, t.seq_len, t.loc_table.data_ptr(), t.loc_table.numel(),
                                             t.seed_table.data_ptr(), t.seed_table.shape[0], C.byref(self.h)),
                 "bwams_emf_from_device")
        elif path is not None:
            _chk(lib().bwams_emf_open(index.h, path.encode(), C.byref(self.h)), "bwams_emf_open")
        else:
            loc = np.ascontiguousarray(table.loc_table, dtype=np.uint32)
            seeds = np.ascontiguousarray(table.seed_table, dtype=np.uint32)
            _chk(lib().bwams_emf_from_host(index.h, table.seed_len, table.seq_len, _p(loc), len(loc), _p(seeds),
                                           len(seeds), C.byref(self.h)), "bwams_emf_from_host")

    @classmethod
    def build(cls, index: Index, seed_len: int = 150, slack: float = 1.1) -> "Emf":
        """bwams_emf_build: the table from the resident forward reference, on the GPU"""
        self = cls.__new__(cls)
        self.index = index
        self._keep = None
        self.h = C.c_void_p()
        _chk(lib().bwams_emf_build(index.h, seed_len, slack, C.byref(self.h)), "bwams_emf_build")
        return self

    def info(self):
        sl, ne, nl = C.c_int32(0), C.c_uint32(0), C.c_uint32(0)
        nu, nk, ms = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _chk(lib().bwams_emf_info(self.h, C.byref(sl), C.byref(ne), C.byref(nl), C.byref(nu), C.byref(nk), C.byref(ms)), "bwams_emf_info")
        return {"seed_len": sl.value, "num_seed_entry": ne.value, "num_loc_entry": nl.value, "n_used": nu.value, "n_key": nk.value,
                "build_ms": ms.value}

    def fetch_table(self):
        i = self.info()
        loc = np.zeros(max(i["num_loc_entry"], 1), dtype=np.uint32)
        seeds = np.zeros((i["num_seed_entry"], 4), dtype=np.uint32)
        _chk(lib().bwams_emf_table_fetch(self.h, _p(loc), _p(seeds)), "bwams_emf_table_fetch")
        return loc[:i["num_loc_entry"]], seeds

    def save(self, path: str):
        _chk(lib().bwams_emf_save(self.h, path.encode()), "bwams_emf_save")

    def close(self):
        if self.h:
            lib().bwams_emf_close(self.h)
            self.h = None


class Ert:
    """ERT index resident in HBM: the reference's <prefix>.kmer_table / <prefix>.mlt_table, from files or host arrays."""

    def __init__(self, index: Index, kmer_table=None, mlt_table=None, kmer: int = 15, xmer: int = 4, read_len: int = 151,
                 prefix: str | None = None):
        self.index = index
        self.h = C.c_void_p()
        if prefix is not None:
            _chk(lib().bwams_ert_open(index.h, prefix.encode(), read_len, C.byref(self.h)), "bwams_ert_open")
        else:
            kt = np.ascontiguousarray(kmer_table, dtype=np.uint64)
            mt = np.ascontiguousarray(mlt_table, dtype=np.uint8)
            assert len(kt) == 4 ** kmer
            _chk(lib().bwams_ert_from_host(index.h, _p(kt), kmer, xmer, read_len, _p(mt), len(mt), C.byref(self.h)),
                 "bwams_ert_from_host")

    @classmethod
    def build(cls, index: Index, kmer: int = 15, xmer: int = 4, read_len: int = 151, hit_threshold: int = 256) -> "Ert":
        """bwams_ert_build: the tables from the resident FM-index, on the GPU"""
        self = cls.__new__(cls)
        self.index = index
        self.h = C.c_void_p()
        _chk(lib().bwams_ert_build(index.h, kmer, xmer, read_len, hit_threshold, C.byref(self.h)), "bwams_ert_build")
        return self

    def info(self):
        k, x, rl, nb = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int64(0)
        ms = (C.c_float * 3)()
        _chk(lib().bwams_ert_info(self.h, C.byref(k), C.byref(x), C.byref(rl), C.byref(nb), ms), "bwams_ert_info")
        return {"kmer": k.value, "xmer": x.value, "read_len": rl.value, "mlt_bytes": nb.value, "build_ms": list(ms)}

    def fetch(self, pad: int = 0):
        """-> (kmer_table, mlt_table); with pad the returned tree array is a view of a buffer `pad` bytes longer (a host
        reader that loads 5- / 8-byte fields near the end needs no copy of a 48 GB array to be safe)"""
        i = self.info()
        kt = np.zeros(4 ** i["kmer"], dtype=np.uint64)
        mt = np.zeros(max(i["mlt_bytes"], 1) + pad, dtype=np.uint8)
        _chk(lib().bwams_ert_fetch(self.h, _p(kt), _p(mt)), "bwams_ert_fetch")
        return kt, mt[:i["mlt_bytes"]]

    def save(self, prefix: str):
        _chk(lib().bwams_ert_save(self.h, prefix.encode()), "bwams_ert_save")

    def nbytes(self) -> int:
        return int(lib().bwams_ert_bytes(self.h))

    def set_fat(self, on: bool):
        """keep (derive again) or give back the walk's resident entry + tree-head table (bwams_ert_set_fat)"""
        _chk(lib().bwams_ert_set_fat(self.h, 1 if on else 0), "bwams_ert_set_fat")

    def close(self):
        if self.h:
            lib().bwams_ert_close(self.h)
            self.h = None


class Batch:
    def __init__(self, index: Index, max_reads: int, max_bases: int, max_smem: int = 0, max_sa: int = 0):
        self.index = index
        self.h = C.c_void_p()
        _chk(lib().bwams_batch_create(index.h, max_reads, max_bases, max_smem, max_sa, C.byref(self.h)),
             "bwams_batch_create")
        self.max_reads = max_reads
        self.max_smem = max_smem if max_smem > 0 else 24 * max_reads + 1024
        self.max_sa = max_sa if max_sa > 0 else 64 * max_reads + 1024

    def seed(self, enc, cum, opt: SeedOpt | None = None, skip=None, with_sa: bool = True):
        """Seeding on host buffers: (smems, sa_coord, sa_off).  Upload, run, then download into arrays of
        exactly the produced size (the one-call C entry point needs caller buffers of capacity size)."""
        opt = opt or default_seed_opt()
        self.seed_upload(enc, cum, skip)
        self.seed_run(opt, with_sa)
        sm, coord, off = self.seed_fetch()
        if with_sa:
            return sm, coord, off
        return sm, None, None

    def seed_onecall(self, enc, cum, opt: SeedOpt | None = None, skip=None, with_sa: bool = True):
        """bwams_seed_fmi with capacity-sized caller buffers (what a reference-side caller does)."""
        opt = opt or default_seed_opt()
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        cum = np.ascontiguousarray(cum, dtype=np.int64)
        sk = np.ascontiguousarray(skip, dtype=np.uint8) if skip is not None else None
        nseq = len(cum) - 1
        sm = np.empty(self.max_smem, dtype=SMEM_DTYPE)
        ns, na = C.c_int64(0), C.c_int64(0)
        coord = np.empty(self.max_sa, dtype=np.int64) if with_sa else None
        off = np.empty(self.max_smem + 1, dtype=np.int64) if with_sa else None
        _chk(lib().bwams_seed_fmi(self.h, _p(enc), _p(cum), _p(sk), nseq, C.byref(opt), _p(sm), self.max_smem,
                                  C.byref(ns), _p(coord), self.max_sa, _p(off), C.byref(na)), "bwams_seed_fmi")
        n = ns.value
        if with_sa:
            return sm[:n].copy(), coord[:na.value].copy(), off[:n + 1].copy()
        return sm[:n].copy(), None, None

    # resident form
    def seed_upload(self, enc, cum, skip=None):
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        cum = np.ascontiguousarray(cum, dtype=np.int64)
        sk = np.ascontiguousarray(skip, dtype=np.uint8) if skip is not None else None
        _chk(lib().bwams_seed_upload(self.h, _p(enc), _p(cum), _p(sk), len(cum) - 1), "bwams_seed_upload")
        self._nseq = len(cum) - 1

    def seed_upload_device(self, enc_ptr: int, cum, skip=None):
        """Same, with the read bytes already in this GPU's memory (enc_ptr = device address of the chunk's first base)."""
        cum = np.ascontiguousarray(cum, dtype=np.int64)
        sk = np.ascontiguousarray(skip, dtype=np.uint8) if skip is not None else None
        _chk(lib().bwams_seed_upload(self.h, C.c_void_p(enc_ptr), _p(cum), _p(sk), len(cum) - 1), "bwams_seed_upload")
        self._nseq = len(cum) - 1

    def seed_run(self, opt: SeedOpt | None = None, with_sa: bool = True):
        opt = opt or default_seed_opt()
        _chk(lib().bwams_seed_run(self.h, C.byref(opt), 1 if with_sa else 0), "bwams_seed_run")

    def seed_run_ert(self, ert: "Ert", opt: SeedOpt | None = None, with_sa: bool = True):
        """Seeding over the ERT: same outputs as seed_run (seed_fetch, chain_run follow unchanged)."""
        opt = opt or default_seed_opt()
        _chk(lib().bwams_seed_run_ert(self.h, ert.h, C.byref(opt), 1 if with_sa else 0), "bwams_seed_run_ert")

    def seed_counts(self):
        ns, na = C.c_int64(0), C.c_int64(0)
        _chk(lib().bwams_seed_counts(self.h, C.byref(ns), C.byref(na)), "bwams_seed_counts")
        return ns.value, na.value

    def seed_fetch(self):
        ns, na = self.seed_counts()
        sm = np.zeros(max(ns, 1), dtype=SMEM_DTYPE)
        coord = np.zeros(max(na, 1), dtype=np.int64)
        off = np.zeros(ns + 1, dtype=np.int64)
        _chk(lib().bwams_seed_fetch(self.h, _p(sm), len(sm), _p(coord), len(coord), _p(off)), "bwams_seed_fetch")
        return sm[:ns], coord[:na], off

    def bsw(self, pairs, ref, qer, w: int, opt: SwOpt | None = None):
        opt = opt or default_sw_opt()
        p = np.ascontiguousarray(pairs).copy()
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        qer = np.ascontiguousarray(qer, dtype=np.uint8)
        _chk(lib().bwams_bsw_extend(self.h, _p(p), len(p), _p(ref), len(ref), _p(qer), len(qer), w, C.byref(opt)),
             "bwams_bsw_extend")
        return p

    def bsw_upload(self, pairs, ref, qer):
        p = np.ascontiguousarray(pairs)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        qer = np.ascontiguousarray(qer, dtype=np.uint8)
        _chk(lib().bwams_bsw_upload(self.h, _p(p), len(p), _p(ref), len(ref), _p(qer), len(qer)), "bwams_bsw_upload")
        self._n_pairs = len(p)

    def bsw_run(self, w: int, opt: SwOpt | None = None):
        opt = opt or default_sw_opt()
        _chk(lib().bwams_bsw_run(self.h, w, C.byref(opt)), "bwams_bsw_run")

    def bsw_fetch(self):
        p = np.zeros(self._n_pairs, dtype=SEQPAIR_DTYPE)
        _chk(lib().bwams_bsw_fetch(self.h, _p(p), len(p)), "bwams_bsw_fetch")
        return p

    def chain_run(self, opt: MemOpt | None = None):
        """mem_chain_seeds + mem_chain_flt over the resident seeds -> (n_chains, n_seeds)."""
        opt = opt or default_mem_opt()
        nc, ns = C.c_int64(0), C.c_int64(0)
        _chk(lib().bwams_chain_run(self.h, C.byref(opt), C.byref(nc), C.byref(ns)), "bwams_chain_run")
        self._n_chain = (nc.value, ns.value)
        return self._n_chain

    def chain_fetch(self):
        nc, ns = self._n_chain
        chains = np.zeros(nc, CHAIN_DTYPE)
        seeds = np.zeros(ns, CHAIN_SEED_DTYPE)
        off = np.zeros(self._nseq + 1, np.int64)
        _chk(lib().bwams_chain_fetch(self.h, _p(chains), nc, _p(seeds), ns, _p(off)), "bwams_chain_fetch")
        return chains, seeds, off

    def chain_upload(self, chains, seeds, chain_off):
        chains = np.ascontiguousarray(chains, dtype=CHAIN_DTYPE)
        seeds = np.ascontiguousarray(seeds, dtype=CHAIN_SEED_DTYPE)
        chain_off = np.ascontiguousarray(chain_off, np.int64)
        _chk(lib().bwams_chain_upload(self.h, _p(chains), len(chains), _p(seeds), len(seeds), _p(chain_off)),
             "bwams_chain_upload")
        self._n_chain = (len(chains), len(seeds))

    def extend_build(self, opt: MemOpt | None = None):
        opt = opt or default_mem_opt()
        nl, nr = C.c_int64(0), C.c_int64(0)
        _chk(lib().bwams_extend_build(self.h, C.byref(opt), C.byref(nl), C.byref(nr)), "bwams_extend_build")
        return nl.value, nr.value

    def extend_run(self, opt: MemOpt | None = None) -> int:
        opt = opt or default_mem_opt()
        n = C.c_int64(0)
        _chk(lib().bwams_extend_run(self.h, C.byref(opt), C.byref(n)), "bwams_extend_run")
        return n.value

    def extend_fetch(self):
        """-> (regs, reg_off, seed_aln)"""
        n = self._n_chain[1]
        regs = np.zeros(n, ALNREG_DTYPE)
        off = np.zeros(self._nseq + 1, np.int64)
        aln = np.zeros(n, np.int32)
        _chk(lib().bwams_extend_fetch(self.h, _p(regs), n, _p(off), _p(aln)), "bwams_extend_fetch")
        return regs, off, aln

    def dedup_run(self, opt: MemOpt | None = None) -> int:
        """The tail of mem_kernel2_core (mem_sort_dedup_patch ...) over the regions of extend_run."""
        opt = opt or default_mem_opt()
        n = C.c_int64(0)
        _chk(lib().bwams_dedup_run(self.h, C.byref(opt), C.byref(n)), "bwams_dedup_run")
        self._n_final = n.value
        return n.value

    def dedup_fetch(self):
        regs = np.zeros(self._n_final, ALNREG_DTYPE)
        off = np.zeros(self._nseq + 1, np.int64)
        _chk(lib().bwams_dedup_fetch(self.h, _p(regs), self._n_final, _p(off)), "bwams_dedup_fetch")
        return regs, off

    def emf_regs(self, emf: "Emf", opt: MemOpt | None = None):
        """mem_perfect2reg for the reads the last emf_run resolved -> (regs, reg_off, first_is_rev)."""
        opt = opt or default_mem_opt()
        n = C.c_int64(0)
        _chk(lib().bwams_emf_regs_run(self.h, emf.h, C.byref(opt), C.byref(n)), "bwams_emf_regs_run")
        regs = np.zeros(n.value, ALNREG_DTYPE)
        off = np.zeros(self._nseq + 1, np.int64)
        rev = np.zeros(self._nseq, np.uint8)
        _chk(lib().bwams_emf_regs_fetch(self.h, _p(regs), n.value, _p(off), _p(rev)), "bwams_emf_regs_fetch")
        return regs, off, rev

    def pestat(self, opt: MemOpt | None = None):
        """mem_pestat over the final regions (reads 2i, 2i+1 = pair i) -> 4 records FF, FR, RF, RR."""
        opt = opt or default_mem_opt()
        pes = np.zeros(4, PESTAT_DTYPE)
        _chk(lib().bwams_pestat(self.h, C.byref(opt), _p(pes)), "bwams_pestat")
        return pes

    def chain_run_ert(self, mems, mem_off, hits, hit_off, opt: MemOpt | None = None):
        """ERT mode: chain the MEMs / hits of the reference's ERT walk (reads uploaded with seed_upload)
        -> (chains, chain seeds)."""
        opt = opt or default_mem_opt()
        mems = np.ascontiguousarray(mems, ERT_MEM_DTYPE)
        mem_off = np.ascontiguousarray(mem_off, np.int64)
        hits = np.ascontiguousarray(hits, np.uint64)
        hit_off = np.ascontiguousarray(hit_off, np.int64)
        nc, ns = C.c_int64(0), C.c_int64(0)
        _chk(lib().bwams_chain_run_ert(self.h, C.byref(opt), _p(mems), _p(mem_off), _p(hits), _p(hit_off), C.byref(nc), C.byref(ns)),
             "bwams_chain_run_ert")
        self._n_chain = (nc.value, ns.value)
        return self._n_chain

    def reg2aln(self, opt: MemOpt | None = None, source: int = 0):
        """mem_reg2aln over the final regions (source 0: after dedup_run; 1: after pair_run): (records, CIGAR pool, MD pool)."""
        opt = opt or default_mem_opt()
        n, nc, nm = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _chk(lib().bwams_reg2aln_run(self.h, C.byref(opt), source, C.byref(n), C.byref(nc), C.byref(nm)), "bwams_reg2aln_run")
        aln = np.zeros(max(n.value, 1), ALN_DTYPE)
        cig = np.zeros(max(nc.value, 1), np.uint32)
        md = np.zeros(max(nm.value, 1), np.uint8)
        _chk(lib().bwams_reg2aln_fetch(self.h, _p(aln), len(aln), _p(cig), len(cig), _p(md), len(md)), "bwams_reg2aln_fetch")
        return aln[:n.value], cig[:nc.value], md[:nm.value]

    def sam_upload(self, names, quals=None, comments=None):
        """Names (list of bytes), qualities (uint8 array laid out like the reads, or None), comments (list of bytes / None, or
        None) of the uploaded chunk: what mem_aln2sam prints besides the alignment."""
        nb = b"".join(names)
        noff = np.zeros(len(names) + 1, np.int64)
        noff[1:] = np.cumsum([len(x) for x in names])
        cb, coff = None, None
        if comments is not None:
            cs = [c or b"" for c in comments]
            cb = b"".join(cs)
            coff = np.zeros(len(cs) + 1, np.int64)
            coff[1:] = np.cumsum([len(x) for x in cs])
        q = np.ascontiguousarray(quals, np.uint8) if quals is not None else None
        _chk(lib().bwams_sam_upload(self.h, nb, _p(noff), _p(q) if q is not None else None, cb, _p(coff) if coff is not None else None),
             "bwams_sam_upload")

    def sam_run(self, opt: MemOpt | None = None, sopt=None) -> int:
        """mem_reg2sam of every read (single-end; after mark_primary_se and reg2aln(source=1)) -> bytes of SAM text."""
        opt = opt or default_mem_opt()
        sopt = sopt or default_sam_opt()
        n = C.c_int64(0)
        _chk(lib().bwams_sam_run(self.h, C.byref(opt), C.byref(sopt), C.byref(n)), "bwams_sam_run")
        self._sam_bytes = n.value
        return n.value

    def process_chunk(self, fastq, paired: bool = False, emf=None, ert=None, seed_opt=None, opt: MemOpt | None = None, sopt=None,
                      pes=None, n_processed: int = 0, flags: int = 0, copy_comment: bool = False, fetch: bool = True):
        """mem_process_seqs for one chunk, text to text (bwams_process_chunk) -> (SAM text, read_off), or the byte count with
        fetch=False.  fastq: bytes, or (device address, n_bytes) for text already in this GPU's memory."""
        seed_opt = seed_opt or default_seed_opt()
        opt = opt or default_mem_opt()
        sopt = sopt or default_sam_opt()
        n, nb = C.c_int64(0), C.c_int64(0)
        pp = _p(np.ascontiguousarray(pes, PESTAT_DTYPE)) if pes is not None else None
        src, nbytes = (C.c_void_p(fastq[0]), fastq[1]) if isinstance(fastq, tuple) else (fastq, len(fastq))
        _chk(lib().bwams_process_chunk(self.h, emf.h if emf is not None else None, ert.h if ert is not None else None, C.byref(seed_opt),
                                       C.byref(opt), C.byref(sopt), src, C.c_int64(nbytes), 1 if paired else 0, pp,
                                       C.c_int64(n_processed), flags | (0x100 if copy_comment else 0), C.byref(n), C.byref(nb)), "bwams_process_chunk")
        self._nseq, self._sam_bytes = n.value, nb.value
        if not fetch:
            return nb.value
        text, off, _ = self.sam_fetch()
        return text, off

    def process_reads(self, enc, cum, names, name_off, quals=None, comments=None, comment_off=None, paired: bool = False, emf=None,
                      ert=None, seed_opt=None, opt: MemOpt | None = None, sopt=None, pes=None, n_processed: int = 0, flags: int = 0,
                      fetch: bool = True):
        """mem_process_seqs for a chunk of parsed records (bwams_process_reads: the form the reference hands over) -> (SAM text,
        read_off), or the byte count with fetch=False."""
        seed_opt = seed_opt or default_seed_opt()
        opt = opt or default_mem_opt()
        sopt = sopt or default_sam_opt()
        a = _flat_records(enc, cum, names, name_off, quals, comments, comment_off)
        nb = C.c_int64(0)
        pp = _p(np.ascontiguousarray(pes, PESTAT_DTYPE)) if pes is not None else None
        _chk(lib().bwams_process_reads(self.h, emf.h if emf is not None else None, ert.h if ert is not None else None, C.byref(seed_opt),
                                       C.byref(opt), C.byref(sopt), *a, 1 if paired else 0, pp, C.c_int64(n_processed), flags, C.byref(nb)),
             "bwams_process_reads")
        self._nseq, self._sam_bytes = len(cum) - 1, nb.value
        if not fetch:
            return nb.value
        text, off, _ = self.sam_fetch()
        return text, off

    def process_chunk2(self, fastq1: bytes, fastq2: bytes, emf=None, ert=None, seed_opt=None, opt: MemOpt | None = None, sopt=None, pes=None,
                       n_processed: int = 0, flags: int = 0, copy_comment: bool = False):
        """A paired-end chunk given as the two files' texts (bwams_process_chunk2) -> (SAM text, read_off)."""
        seed_opt = seed_opt or default_seed_opt()
        opt = opt or default_mem_opt()
        sopt = sopt or default_sam_opt()
        n, nb = C.c_int64(0), C.c_int64(0)
        pp = _p(np.ascontiguousarray(pes, PESTAT_DTYPE)) if pes is not None else None
        _chk(lib().bwams_process_chunk2(self.h, emf.h if emf is not None else None, ert.h if ert is not None else None, C.byref(seed_opt),
                                        C.byref(opt), C.byref(sopt), fastq1, C.c_int64(len(fastq1)), fastq2, C.c_int64(len(fastq2)), pp,
                                        C.c_int64(n_processed), flags | (0x100 if copy_comment else 0), C.byref(n), C.byref(nb)),
             "bwams_process_chunk2")
        self._nseq, self._sam_bytes = n.value, nb.value
        text, off, _ = self.sam_fetch()
        return text, off

    def process_chunk_smart(self, fastq, emf=None, ert=None, seed_opt=None, opt: MemOpt | None = None, sopt=None, pes=None,
                            n_processed: int = 0, flags: int = 0, copy_comment: bool = False):
        """process()'s MEM_F_SMARTPE branch for one chunk (bwams_process_chunk_smart): single reads and interleaved pairs mixed
        -> (SAM text, read_off, number of single reads)."""
        seed_opt = seed_opt or default_seed_opt()
        opt = opt or default_mem_opt()
        sopt = sopt or default_sam_opt()
        n, n0, nb = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        pp = _p(np.ascontiguousarray(pes, PESTAT_DTYPE)) if pes is not None else None
        src, nbytes = (C.c_void_p(fastq[0]), fastq[1]) if isinstance(fastq, tuple) else (fastq, len(fastq))
        _chk(lib().bwams_process_chunk_smart(self.h, emf.h if emf is not None else None, ert.h if ert is not None else None, C.byref(seed_opt),
                                             C.byref(opt), C.byref(sopt), src, C.c_int64(nbytes), pp, C.c_int64(n_processed),
                                             flags | (0x100 if copy_comment else 0), C.byref(n), C.byref(n0), C.byref(nb)),
             "bwams_process_chunk_smart")
        self._nseq, self._sam_bytes = n.value, nb.value
        text, off, _ = self.sam_fetch()
        return text, off, n0.value

    def sam_run_emf(self, emf: "Emf", opt: MemOpt | None = None, sopt=None) -> int:
        """sam_run for a chunk that went through emf_run + emf_regs_run: resolved reads get mem_perfect2sam_cont's records."""
        opt = opt or default_mem_opt()
        sopt = sopt or default_sam_opt()
        n = C.c_int64(0)
        _chk(lib().bwams_sam_run_emf(self.h, C.byref(opt), C.byref(sopt), emf.h, C.byref(n)), "bwams_sam_run_emf")
        self._sam_bytes = n.value
        return n.value

    def sam_run_pe(self, pes, opt: MemOpt | None = None, sopt=None) -> int:
        """mem_sam_pe's text of every pair (after pair_run and reg2aln(source=1)) -> bytes of SAM text."""
        opt = opt or default_mem_opt()
        sopt = sopt or default_sam_opt()
        pes = np.ascontiguousarray(pes, PESTAT_DTYPE)
        n = C.c_int64(0)
        _chk(lib().bwams_sam_run_pe(self.h, C.byref(opt), C.byref(sopt), _p(pes), C.byref(n)), "bwams_sam_run_pe")
        self._sam_bytes = n.value
        return n.value

    def sam_fetch(self, n_regs: int = 0):
        """(SAM text as bytes, read_off[nseq + 1], device-side mapq per region)."""
        buf = np.zeros(max(self._sam_bytes, 1), np.uint8)
        off = np.zeros(self._nseq + 1, np.int64)
        mq = np.zeros(max(n_regs, 1), np.int32)
        _chk(lib().bwams_sam_fetch(self.h, _p(buf), len(buf), _p(off), _p(mq) if n_regs else None, len(mq)), "bwams_sam_fetch")
        return bytes(buf[:self._sam_bytes]), off, mq[:n_regs]

    def reg2aln_sam(self, opt: MemOpt | None = None, sopt=None, pes=None, fetch: bool = True):
        """mem_reg2aln of the regions the SAM text needs only (after mark_primary_se, or pair_run with its pes) ->
        (records, CIGAR pool, MD pool, number of regions aligned); skipped regions hold an unmapped record."""
        opt = opt or default_mem_opt()
        sopt = sopt or default_sam_opt()
        n, nn, nc, nm = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        pp = _p(np.ascontiguousarray(pes, PESTAT_DTYPE)) if pes is not None else None
        _chk(lib().bwams_reg2aln_run_sam(self.h, C.byref(opt), C.byref(sopt), pp, C.byref(n), C.byref(nn), C.byref(nc), C.byref(nm)),
             "bwams_reg2aln_run_sam")
        if not fetch:
            return n.value, nn.value
        aln = np.zeros(max(n.value, 1), ALN_DTYPE)
        cig = np.zeros(max(nc.value, 1), np.uint32)
        md = np.zeros(max(nm.value, 1), np.uint8)
        _chk(lib().bwams_reg2aln_fetch(self.h, _p(aln), len(aln), _p(cig), len(cig), _p(md), len(md)), "bwams_reg2aln_fetch")
        return aln[:n.value], cig[:nc.value], md[:nm.value], nn.value

    def pestat_keys(self, opt: MemOpt | None = None) -> np.ndarray:
        """One key per qualifying pair of this batch (orientation << 60 | insert size), sorted."""
        opt = opt or default_mem_opt()
        keys = np.zeros(max(self._nseq // 2, 1), np.uint64)
        n = C.c_int64(0)
        _chk(lib().bwams_pestat_keys(self.h, C.byref(opt), _p(keys), len(keys), C.byref(n)), "bwams_pestat_keys")
        return keys[:n.value].copy()

    def mark_primary_se(self, opt: MemOpt | None = None, id_base: int = 0, sopt=None):
        """Single-end chunk: mem_mark_primary_se of every read's final regions (then pair_fetch / reg2aln(source=1)); sopt carries
        MEM_F_PRIMARY5 and its T (bwams_pair_run_sam)."""
        opt = opt or default_mem_opt()
        n, nt = C.c_int64(0), C.c_int64(0)
        _chk(lib().bwams_pair_run_sam(self.h, C.byref(opt), C.byref(sopt) if sopt is not None else None, None, C.c_int64(id_base), 4,
                                      C.byref(n), C.byref(nt)), "bwams_pair_run_sam")
        self._n_pair_regs = n.value
        return n.value

    def pair_run(self, pes, opt: MemOpt | None = None, id_base: int = 0, no_rescue: bool = False, use_ert: bool = False, sopt=None):
        """Mate rescue + mem_mark_primary_se + mem_pair over the final regions (reads 2p, 2p+1 = pair p)
        -> (regions, rescue alignments).  sopt: MEM_F_PRIMARY5 / MEM_F_NOPAIRING / MEM_F_NO_RESCUE of its flag (bwams_pair_run_sam)."""
        opt = opt or default_mem_opt()
        pes = np.ascontiguousarray(pes, PESTAT_DTYPE)
        n, nt = C.c_int64(0), C.c_int64(0)
        _chk(lib().bwams_pair_run_sam(self.h, C.byref(opt), C.byref(sopt) if sopt is not None else None, _p(pes), C.c_int64(id_base),
                                      int(no_rescue) | (int(use_ert) << 1), C.byref(n), C.byref(nt)), "bwams_pair_run_sam")
        self._n_pair_regs = n.value
        return n.value, nt.value

    def pair_fetch(self):
        """-> (regs, reg_off, pairs) as mem_sam_pe_batch_post holds them before its MAPQ / SAM part."""
        regs = np.zeros(self._n_pair_regs, ALNREG_DTYPE)
        off = np.zeros(self._nseq + 1, np.int64)
        pairs = np.zeros(self._nseq // 2, PAIR_DTYPE)
        _chk(lib().bwams_pair_fetch(self.h, _p(regs), len(regs), _p(off), _p(pairs)), "bwams_pair_fetch")
        return regs, off, pairs

    def extend_tasks_fetch(self, side: int):
        n, rb, qb = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        rc = lib().bwams_extend_tasks_fetch(self.h, side, None, 0, None, 0, None, 0, C.byref(n), C.byref(rb), C.byref(qb))
        if rc not in (0, -4):
            _chk(rc, "bwams_extend_tasks_fetch")
        pairs = np.zeros(n.value, SEQPAIR_DTYPE)
        ref = np.zeros(rb.value, np.uint8)
        qer = np.zeros(qb.value, np.uint8)
        _chk(lib().bwams_extend_tasks_fetch(self.h, side, _p(pairs), n.value, _p(ref), rb.value, _p(qer), qb.value,
                                            C.byref(n), C.byref(rb), C.byref(qb)), "bwams_extend_tasks_fetch")
        return pairs, ref, qer

    def ksw_align(self, pairs, ref, qer, opt: SwOpt | None = None):
        """Mate-rescue local SW: int32[n, 7] = score, te, qe, score2, te2, tb, qb (pairs[i].h0 = xtra)."""
        opt = opt or default_sw_opt()
        p = np.ascontiguousarray(pairs)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        qer = np.ascontiguousarray(qer, dtype=np.uint8)
        out = np.zeros((len(p), 7), dtype=np.int32)
        _chk(lib().bwams_ksw_align(self.h, _p(p), len(p), _p(ref), len(ref), _p(qer), len(qer), C.byref(opt), _p(out)),
             "bwams_ksw_align")
        return out

    def emf_probe(self, emf: Emf, enc, cum):
        """(perfect uint32[n, 2] = flags, location; code uint8[n])."""
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        cum = np.ascontiguousarray(cum, dtype=np.int64)
        n = len(cum) - 1
        out = np.zeros((max(n, 1), 2), dtype=np.uint32)
        code = np.zeros(max(n, 1), dtype=np.uint8)
        _chk(lib().bwams_emf_probe(self.h, emf.h, _p(enc), _p(cum), n, _p(out), _p(code)), "bwams_emf_probe")
        return out[:n], code[:n]

    def emf_regs_merge(self) -> int:
        """Paired-end: the regions of the EMF-resolved reads join the final regions (after dedup_run and pestat, before pair_run)."""
        n = C.c_int64(0)
        _chk(lib().bwams_emf_regs_merge(self.h, C.byref(n)), "bwams_emf_regs_merge")
        return n.value

    def emf_run(self, emf: Emf):
        _chk(lib().bwams_emf_run(self.h, emf.h), "bwams_emf_run")

    def emf_fetch(self, n: int):
        out = np.zeros((max(n, 1), 2), dtype=np.uint32)
        code = np.zeros(max(n, 1), dtype=np.uint8)
        _chk(lib().bwams_emf_fetch(self.h, _p(out), _p(code)), "bwams_emf_fetch")
        return out[:n], code[:n]

    def stats(self) -> Stats:
        s = Stats()
        _chk(lib().bwams_batch_stats(self.h, C.byref(s)), "bwams_batch_stats")
        return s

    def sync(self):
        _chk(lib().bwams_batch_sync(self.h), "bwams_batch_sync")

    def close(self):
        if self.h:
            lib().bwams_batch_destroy(self.h)
            self.h = None


class _Kept(tuple):
    """an argument tuple that owns the arrays its pointers point into: alive as long as the caller's local holds it"""
    def __new__(cls, args, keep):
        t = super().__new__(cls, args)
        t.keep = keep
        return t


def _flat_records(enc, cum, names, name_off, quals, comments, comment_off):
    """the argument block (enc, cum, n, names, name_off, quals, comments, comment_off) of bwams_process_reads and its relatives;
    the converted arrays live in the returned tuple (hold it in a local until the C call has returned: ctypes releases the GIL)"""
    enc = np.ascontiguousarray(enc, np.uint8)
    cum = np.ascontiguousarray(cum, np.int64)
    names = np.ascontiguousarray(names, np.uint8)
    name_off = np.ascontiguousarray(name_off, np.int64)
    q = np.ascontiguousarray(quals, np.uint8) if quals is not None else None
    c = np.ascontiguousarray(comments, np.uint8) if comments is not None else None
    co = np.ascontiguousarray(comment_off, np.int64) if comment_off is not None else None
    return _Kept((_p(enc), _p(cum), C.c_int64(len(cum) - 1), _p(names), _p(name_off), _p(q), _p(c), _p(co)), (enc, cum, names, name_off, q, c, co))


def debug_reload():
    """re-read the library's debugging aids / A-B switches from the environment (they are parsed once otherwise)"""
    lib().bwams_debug_reload()


def shard_bounds(n_reads: int, n_shards: int, paired: bool = False) -> np.ndarray:
    """bwams_shard_bounds: where the shards of a chunk begin (n_shards + 1 read indices)"""
    b = np.zeros(n_shards + 1, np.int64)
    _chk(lib().bwams_shard_bounds(C.c_int64(n_reads), n_shards, 1 if paired else 0, _p(b)), "bwams_shard_bounds")
    return b


class Multi:
    """One chunk over several batches (one per GPU, or several on one GPU) behind one call: host/chunk_multi.cpp."""

    def __init__(self, batches, emfs=None, erts=None):
        self.batches = list(batches)
        n = len(self.batches)
        arr = (C.c_void_p * n)(*[b.h for b in self.batches])
        ea = (C.c_void_p * n)(*[e.h for e in emfs]) if emfs else None
        ra = (C.c_void_p * n)(*[e.h for e in erts]) if erts else None
        self.h = C.c_void_p()
        _chk(lib().bwams_multi_create(arr, ea, ra, n, C.byref(self.h)), "bwams_multi_create")
        self._n = 0

    def process_reads(self, enc, cum, names, name_off, quals=None, comments=None, comment_off=None, paired: bool = False, seed_opt=None,
                      opt: MemOpt | None = None, sopt=None, pes=None, n_processed: int = 0, flags: int = 0):
        """-> (SAM text of the chunk in read order, read_off[n + 1])"""
        seed_opt = seed_opt or default_seed_opt()
        opt = opt or default_mem_opt()
        sopt = sopt or default_sam_opt()
        a = _flat_records(enc, cum, names, name_off, quals, comments, comment_off)
        nb = C.c_int64(0)
        pp = _p(np.ascontiguousarray(pes, PESTAT_DTYPE)) if pes is not None else None
        L = lib()
        L.bwams_multi_error.restype = C.c_char_p
        rc = L.bwams_multi_process_reads(self.h, C.byref(seed_opt), C.byref(opt), C.byref(sopt), *a, 1 if paired else 0, pp,
                                         C.c_int64(n_processed), flags, C.byref(nb))
        if rc:
            raise BwamsError(rc, "bwams_multi_process_reads", (L.bwams_multi_error(self.h) or b"").decode())
        n = len(cum) - 1
        buf = np.zeros(max(nb.value, 1), np.uint8)
        off = np.zeros(n + 1, np.int64)
        _chk(L.bwams_multi_fetch(self.h, _p(buf), C.c_int64(nb.value), _p(off)), "bwams_multi_fetch")
        return buf[:nb.value].tobytes(), off

    def close(self):
        if self.h:
            lib().bwams_multi_destroy(self.h)
            self.h = None


_PINNED = {}          # address of a pinned_array's memory -> its finalizer (pinned_free runs it early)


def pinned_array(n: int, dtype=np.uint8) -> np.ndarray:
    """a numpy view of page-locked host memory (bwams_host_alloc).  Free it with pinned_free(arr) when done; an array that is simply
    dropped is freed when it is garbage collected (a finalizer that may run late, at interpreter exit)"""
    dt = np.dtype(dtype)
    p = C.c_void_p()
    _chk(lib().bwams_host_alloc(C.c_size_t(max(n, 1) * dt.itemsize), C.byref(p)), "bwams_host_alloc")
    buf = (C.c_uint8 * (max(n, 1) * dt.itemsize)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dt, count=n)
    import weakref
    _PINNED[p.value] = weakref.finalize(buf, lambda a=p.value: lib().bwams_host_free(C.c_void_p(a)))
    return arr


def pinned_free(arr: np.ndarray) -> None:
    """give a pinned_array's memory back now (the array must not be used afterwards)"""
    fin = _PINNED.pop(arr.ctypes.data, None)
    if fin is not None:
        fin()


# ---------------------------------------------------------------------------------------------------------------------------------
# The compiled outer boundary (bwa-mem-scale_amd/host/mem_process_seqs_hip.cpp) driven from Python: the reference's own records
# (bseq1_t, mem_opt_t: the layout mirrors of host/bwamem_hip.h) through mem_process_seqs() and the pipeline's two other steps.
# The host layer has C++ linkage, as mem_process_seqs has in the reference; its symbols are looked up by name in the library.
BSEQ1_DTYPE = np.dtype([("l_seq", "<i4"), ("id", "<i4"), ("strbuf", "<u8"), ("name", "<u8"), ("comment", "<u8"), ("seq", "<u8"),
                        ("qual", "<u8"), ("sam", "<u8"), ("perfect", "<u8")])
assert BSEQ1_DTYPE.itemsize == 64
WORK_ITEM = 512          # BATCH_SIZE, src/macro.h:63: reads per seqs[].sam string


class MemOptT(C.Structure):
    """mem_opt_t (src/bwamem.h:89-124, built without AFF): 176 bytes"""
    _fields_ = [("a", C.c_int), ("b", C.c_int), ("o_del", C.c_int), ("e_del", C.c_int), ("o_ins", C.c_int), ("e_ins", C.c_int),
                ("pen_unpaired", C.c_int), ("pen_clip5", C.c_int), ("pen_clip3", C.c_int), ("w", C.c_int), ("zdrop", C.c_int),
                ("max_mem_intv", C.c_uint64), ("T", C.c_int), ("flag", C.c_int), ("min_seed_len", C.c_int), ("min_chain_weight", C.c_int),
                ("max_chain_extend", C.c_int), ("split_factor", C.c_float), ("split_width", C.c_int), ("max_occ", C.c_int),
                ("max_chain_gap", C.c_int), ("n_threads", C.c_int), ("chunk_size", C.c_int64), ("mask_level", C.c_float),
                ("drop_ratio", C.c_float), ("XA_drop_ratio", C.c_float), ("mask_level_redun", C.c_float), ("mapQ_coef_len", C.c_float),
                ("mapQ_coef_fac", C.c_int), ("max_ins", C.c_int), ("max_matesw", C.c_int), ("max_XA_hits", C.c_int),
                ("max_XA_hits_alt", C.c_int), ("mat", C.c_int8 * 25)]


assert C.sizeof(MemOptT) == 176


def mem_opt_init(paired: bool = False) -> MemOptT:
    """mem_opt_init (src/bwamem.cpp:135-171) + bwa_fill_scmat"""
    import math
    o = MemOptT()
    o.a, o.b, o.o_del, o.o_ins, o.e_del, o.e_ins, o.w, o.T, o.zdrop = 1, 4, 6, 6, 1, 1, 100, 30, 100
    o.pen_unpaired, o.pen_clip5, o.pen_clip3, o.max_mem_intv, o.min_seed_len, o.split_width = 17, 5, 5, 20, 19, 10
    o.max_occ, o.max_chain_gap, o.max_ins, o.mask_level, o.drop_ratio = 500, 10000, 10000, 0.50, 0.50
    o.XA_drop_ratio, o.split_factor, o.chunk_size, o.n_threads, o.max_XA_hits, o.max_XA_hits_alt = 0.80, 1.5, 10000000, 1, 5, 200
    o.max_matesw, o.mask_level_redun, o.min_chain_weight, o.max_chain_extend, o.mapQ_coef_len = 50, 0.95, 0, 1 << 30, 50
    o.mapQ_coef_fac = int(math.log(o.mapQ_coef_len))
    k = 0
    for i in range(4):
        for j in range(4):
            o.mat[k] = o.a if i == j else -o.b
            k += 1
        o.mat[k] = -1
        k += 1
    for _ in range(5):
        o.mat[k] = -1
        k += 1
    if paired:
        o.flag |= 0x2
    return o


_host_syms = None


def _host_sym(name: str):
    """the host layer's function `name` (C++ linkage: looked up among the library's dynamic symbols by its mangled prefix)"""
    global _host_syms
    if _host_syms is None:
        out = subprocess.check_output(["nm", "-D", "--defined-only", LIB_PATH]).decode()
        _host_syms = [ln.split()[-1] for ln in out.splitlines() if " T _Z" in ln]
    key = f"_Z{len(name)}{name}"
    hit = [m for m in _host_syms if m.startswith(key) and (len(m) == len(key) or not m[len(key)].islower())]
    if len(hit) != 1:
        raise BwamsError(-3, name, f"host-layer symbol not found or ambiguous: {hit}")
    f = getattr(lib(), hit[0])
    f.restype = C.c_int
    return f


class Seqs:
    """A chunk as the reference holds it between kt_pipeline's steps: a bseq1_t array whose strings live in this object's buffers."""

    def __init__(self, reads: np.ndarray, first_id: int = 0, name_fmt: bytes = b"r%08d", quals: bool = True, name_ids=None):
        """reads: (n, RL) base codes 0..4; names are name_fmt % (first_id + i), or % name_ids[i] (paired-end: the pair's number)"""
        n, RL = reads.shape
        self.n = n
        self.seq = np.zeros((n, RL + 1), np.uint8)
        self.seq[:, :RL] = np.frombuffer(b"ACGTN", np.uint8)[reads]
        w = len(name_fmt % 0) + 1
        self.names = np.zeros((n, w), np.uint8)
        ids = first_id + np.arange(n, dtype=np.int64) if name_ids is None else np.asarray(name_ids, np.int64)
        head = name_fmt.split(b"%")[0]
        nd = w - 1 - len(head)
        self.names[:, :len(head)] = np.frombuffer(head, np.uint8)
        for d in range(nd):
            self.names[:, len(head) + d] = ord("0") + (ids // 10 ** (nd - 1 - d)) % 10
        self.qual = None
        if quals:
            self.qual = np.zeros((n, RL + 1), np.uint8)
            self.qual[:, :RL] = ord("I")
        self.arr = np.zeros(n, BSEQ1_DTYPE)
        self.arr["l_seq"] = RL
        self.arr["id"] = ids.astype(np.int32)
        self.arr["name"] = self.names.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(w)
        self.arr["seq"] = self.seq.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(RL + 1)
        if quals:
            self.arr["qual"] = self.qual.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(RL + 1)

    @property
    def ptr(self):
        return C.c_void_p(self.arr.ctypes.data)

    def drop_sam(self) -> int:
        """free the work items' strings unread (a writer that costs nothing); returns their total length"""
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        libc.strlen.argtypes = [C.c_void_p]
        libc.strlen.restype = C.c_size_t
        total = 0
        for i in range(0, self.n, WORK_ITEM):
            p = int(self.arr["sam"][i])
            if not p:
                raise BwamsError(-3, "drop_sam", f"work item {i // WORK_ITEM} left no text")
            total += libc.strlen(C.c_void_p(p))
            libc.free(C.c_void_p(p))
            self.arr["sam"][i] = 0
        return total

    def take_sam(self) -> bytes:
        """the chunk's SAM text (the work items' strings joined, as step 2 writes them), the strings freed"""
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        parts = []
        for i in range(0, self.n, WORK_ITEM):
            p = int(self.arr["sam"][i])
            if not p:
                raise BwamsError(-3, "take_sam", f"work item {i // WORK_ITEM} left no text")
            parts.append(C.string_at(p))
            libc.free(C.c_void_p(p))
            self.arr["sam"][i] = 0
        return b"".join(parts)


class Worker:
    """bwams_worker (host/bwamem_hip.h): the resident index set of every device and `depth` chunks in flight."""

    def __init__(self, indexes, max_reads: int, max_bases: int, emfs=None, erts=None, depth: int = 1, rg_id: bytes = b""):
        n = len(indexes)
        ia = (C.c_void_p * n)(*[x.h for x in indexes])
        ea = (C.c_void_p * n)(*[e.h for e in emfs]) if emfs else None
        ra = (C.c_void_p * n)(*[e.h for e in erts]) if erts else None
        self.h = C.c_void_p()
        _chk(_host_sym("bwams_worker_create_multi")(ia, ea, ra, n, depth, C.c_int64(max_reads), C.c_int64(max_bases), rg_id, C.byref(self.h)),
             "bwams_worker_create_multi")
        self._keep = (indexes, emfs, erts)

    def _err(self, rc, what):
        if rc:
            f = _host_sym("bwams_worker_error")
            f.restype = C.c_char_p
            raise BwamsError(rc, what, (f(self.h) or b"").decode())

    def set_deferred_collect(self, on: bool):
        _host_sym("bwams_worker_set_deferred_collect")(self.h, 1 if on else 0)

    def stage(self, opt: MemOptT, seqs: Seqs):
        self._err(_host_sym("mem_process_seqs_stage")(C.byref(opt), seqs.n, seqs.ptr, self.h), "mem_process_seqs_stage")

    def process(self, opt: MemOptT, n_processed: int, seqs: Seqs, pes0=None):
        pp = _p(np.ascontiguousarray(pes0, PESTAT_DTYPE)) if pes0 is not None else None
        self._err(_host_sym("bwams_worker_process")(C.byref(opt), C.c_int64(n_processed), seqs.n, seqs.ptr, pp, self.h), "mem_process_seqs")

    def collect(self, opt: MemOptT, seqs: Seqs):
        self._err(_host_sym("mem_process_seqs_collect")(C.byref(opt), seqs.n, seqs.ptr, self.h), "mem_process_seqs_collect")

    def close(self):
        if self.h:
            f = _host_sym("bwams_worker_destroy")
            f.restype = None
            f(self.h)
            self.h = None
