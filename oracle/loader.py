"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Builds oracle/_build/liboracle.so with gcc on first use; loads the real
reference objects from oracle/_ref/ when they exist (built by oracle/Makefile in
the container where /root/reference is mounted; they travel to the GPU box as
prebuilt files).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Smem(C.Structure):
    _fields_ = [("rid", C.c_uint32), ("m", C.c_uint32), ("n", C.c_uint32), ("pad_", C.c_uint32),
                ("k", C.c_int64), ("l", C.c_int64), ("s", C.c_int64)]


SMEM_DTYPE = np.dtype([("rid", "<u4"), ("m", "<u4"), ("n", "<u4"), ("pad_", "<u4"),
                       ("k", "<i8"), ("l", "<i8"), ("s", "<i8")])
SEQPAIR_DTYPE = np.dtype([(n, "<i4") for n in
                          ("idr", "idq", "id", "len1", "len2", "h0", "seqid", "regid",
                           "score", "tle", "gtle", "qle", "gscore", "max_off")])
assert SMEM_DTYPE.itemsize == 40 and SEQPAIR_DTYPE.itemsize == 56


class SeedOpt(C.Structure):
    _fields_ = [("min_seed_len", C.c_int32), ("split_factor", C.c_float),
                ("split_width", C.c_int32), ("max_mem_intv", C.c_int32), ("max_occ", C.c_int32)]


class SwOpt(C.Structure):
    _fields_ = [("o_del", C.c_int32), ("e_del", C.c_int32), ("o_ins", C.c_int32),
                ("e_ins", C.c_int32), ("zdrop", C.c_int32), ("end_bonus", C.c_int32),
                ("mat", C.c_int8 * 25), ("pad_", C.c_int8 * 3)]


class OrcFmi(C.Structure):
    _fields_ = [("ref_seq_len", C.c_int64), ("count", C.c_int64 * 5),
                ("cp_occ", C.c_void_p), ("sa_ms_byte", C.c_void_p), ("sa_ls_word", C.c_void_p),
                ("sentinel_index", C.c_int64),
                ("all_smem", C.c_void_p), ("last_smem", C.c_void_p), ("all_bp", C.c_int32), ("last_bp", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("n_ext", C.c_int64), ("n_ext_blocks", C.c_int64), ("n_sa_lookups", C.c_int64),
                ("n_lf_steps", C.c_int64), ("n_smem", C.c_int64 * 3)]


def default_seed_opt() -> SeedOpt:
    """mem_opt_init defaults (src/bwamem.cpp:135-171)."""
    return SeedOpt(19, 1.5, 10, 20, 500)


def fill_scmat(a: int = 1, b: int = 4):
    """bwa_fill_scmat (src/bwa.cpp:368-377)."""
    m = []
    for i in range(4):
        for j in range(4):
            m.append(a if i == j else -b)
        m.append(-1)
    m += [-1] * 5
    return m


def default_sw_opt(end_bonus: int = 5, a: int = 1, b: int = 4) -> SwOpt:
    o = SwOpt(6, 1, 6, 1, 100, end_bonus)
    for i, v in enumerate(fill_scmat(a, b)):
        o.mat[i] = v
    return o


_lib = None


def build() -> str:
    out = os.path.join(HERE, "_build", "liboracle.so")
    srcs = [os.path.join(HERE, f) for f in ("fmi_oracle.c", "bsw_oracle.c", "ksw_oracle.c", "emf_oracle.c", "chain_oracle.c", "dedup_oracle.c",
                                            "pair_oracle.c", "aln_oracle.c", "ert_oracle.c", "sam_oracle.c", "fastq_oracle.c",
                                            "bwams_oracle.h", "../include/bwams_types.h")]
    if not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", HERE, "_build/liboracle.so"])
    return out


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
        L.orc_fmi_occ.restype = i64
        L.orc_fmi_occ.argtypes = [vp, i64, C.c_int]
        L.orc_backward_ext.restype = None
        L.orc_backward_ext.argtypes = [vp, vp, C.c_int, vp, vp]
        L.orc_collect_smem.restype = i64
        L.orc_collect_smem.argtypes = [vp, vp, vp, vp, vp, i32, vp, i64, vp]
        L.orc_sa_entry.restype = i64
        L.orc_sa_entry.argtypes = [vp, i64, vp]
        L.orc_sa_lookup.restype = i64
        L.orc_sa_lookup.argtypes = [vp, vp, i64, i32, vp, i64, vp, vp]
        L.orc_bsw_scalar.restype = C.c_int
        L.orc_bsw_scalar.argtypes = [vp, C.c_int, vp, C.c_int, vp, i32, C.c_int] + [vp] * 6
        L.orc_bsw_pairs.restype = None
        L.orc_bsw_pairs.argtypes = [vp, vp, vp, vp, i64, i32, vp]
        L.orc_ksw_align2.restype = None
        L.orc_ksw_align2.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp]
        L.orc_kbt_script.restype = i64
        L.orc_kbt_script.argtypes = [i64, vp, vp, vp, vp]
        L.orc_flt_sort.restype = None
        L.orc_flt_sort.argtypes = [i64, vp, vp]
        L.orc_chain_seeds.restype = i64
        L.orc_chain_seeds.argtypes = [vp, vp, vp, i64, vp, vp, vp, i32, C.c_int, vp, i64, vp, i64, vp, vp, vp, vp]
        L.orc_chain2aln.restype = i64
        L.orc_chain2aln.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i64, vp, vp]
        L.orc_task_dump_free.restype = None
        L.orc_task_dump_free.argtypes = [vp]
        L.orc_ksw_global2_score.restype = C.c_int
        L.orc_ksw_global2_score.argtypes = [C.c_int, vp, C.c_int, vp, vp] + [C.c_int] * 5
        L.orc_ars_sort.restype = None
        L.orc_ars_sort.argtypes = [i64, C.c_int, vp, vp, vp, vp]
        L.orc_pestat.restype = None
        L.orc_pestat.argtypes = [vp, i64, C.c_int, vp, vp, vp]
        L.orc_regs_finish.restype = i64
        L.orc_regs_finish.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp]
        _lib = L
    return _lib


KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x10000, 0x20000, 0x40000, 0x80000


def ksw_align2(query, target, xtra: int, opt: SwOpt | None = None):
    """Restated ksw_align2: (score, te, qe, score2, te2, tb, qb)."""
    opt = opt or default_sw_opt()
    q = np.ascontiguousarray(query, dtype=np.uint8)
    t = np.ascontiguousarray(target, dtype=np.uint8)
    out = (C.c_int * 7)()
    lib().orc_ksw_align2(C.byref(opt), len(q), _p(q), len(t), _p(t), xtra, out)
    return tuple(out)


def ref_ksw_align2(L, query, target, xtra: int, opt: SwOpt | None = None):
    """The reference's ksw_align2 from oracle/_ref (it reverses its inputs in place and back)."""
    opt = opt or default_sw_opt()
    q = np.ascontiguousarray(query, dtype=np.uint8).copy()
    t = np.ascontiguousarray(target, dtype=np.uint8).copy()
    out = (C.c_int * 7)()
    L.ref_ksw_align2.restype = None
    L.ref_ksw_align2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.ref_ksw_align2(C.byref(opt), len(q), _p(q), len(t), _p(t), xtra, out)
    return tuple(out)


class OrcEmf(C.Structure):
    _fields_ = [("seed_len", C.c_int32), ("num_loc_entry", C.c_uint32), ("num_seed_entry", C.c_uint32),
                ("seq_len", C.c_uint32), ("loc_table", C.c_void_p), ("seed_table", C.c_void_p), ("ref", C.c_void_p)]


class OracleEMF:
    """EMF table (bwams.emf.EmfTable) + .0123 reference, probed by the restated find_perfect_match_entry."""

    def __init__(self, table, ref_0123):
        self.loc = np.ascontiguousarray(table.loc_table, dtype=np.uint32)
        self.seeds = np.ascontiguousarray(table.seed_table, dtype=np.uint32)
        self.ref = np.ascontiguousarray(ref_0123, dtype=np.uint8)
        self.t = OrcEmf(table.seed_len, len(self.loc), len(self.seeds), table.seq_len,
                        self.loc.ctypes.data, self.seeds.ctypes.data, self.ref.ctypes.data)
        L = lib()
        L.orc_emf_probe.restype = C.c_int
        L.orc_emf_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]

    def probe(self, read):
        r = np.ascontiguousarray(read, dtype=np.uint8)
        fl, loc = C.c_uint32(0), C.c_uint32(0)
        code = lib().orc_emf_probe(C.byref(self.t), _p(r), len(r), C.byref(fl), C.byref(loc))
        return code, fl.value, loc.value

    def probe_many(self, reads):
        out = np.zeros((len(reads), 3), dtype=np.int64)
        for i, r in enumerate(reads):
            out[i] = self.probe(r)
        return out

    def perfect2reg(self, read, flags: int, location: int, l_pac: int, contigs=None, opt=None, cap: int = 1 << 16):
        """Restated mem_perfect2reg for one resolved read -> (regs, first_is_rev)."""
        L = lib()
        L.orc_perfect2reg.restype = C.c_int
        L.orc_perfect2reg.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
        opt = opt or default_mem_opt()
        bns, keep = _bns(l_pac, contigs if contigs is not None else single_contig(l_pac))
        r = np.ascontiguousarray(read, dtype=np.uint8)
        out = np.zeros(cap, ALNREG_DTYPE)
        rev = C.c_int(0)
        n = L.orc_perfect2reg(C.byref(opt), C.byref(self.t), C.byref(bns), _p(r), len(r), flags, location, _p(out), cap, C.byref(rev))
        assert n >= 0
        return out[:n].copy(), rev.value


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleFMI:
    """Holds numpy arrays alive and the orc_fmi_t that points at them."""

    def __init__(self, idx):
        self.cp = np.ascontiguousarray(idx.cp_occ).view(np.uint64)
        self.ms = np.ascontiguousarray(idx.sa_ms_byte)
        self.ls = np.ascontiguousarray(idx.sa_ls_word)
        self.f = OrcFmi(int(idx.ref_seq_len), (C.c_int64 * 5)(*[int(x) for x in idx.count]),
                        self.cp.ctypes.data, self.ms.ctypes.data, self.ls.ctypes.data,
                        int(idx.sentinel_index), None, None, 0, 0)
        self.all_tab = self.last_tab = None

    def build_fma(self, all_bp: int = 11, last_bp: int = 13):
        """Build and attach the FMA tables (all_smem: 128 B x 4^all_bp, last_smem: 16 B x 4^last_bp)."""
        L = lib()
        L.orc_build_all_smem.restype = None
        L.orc_build_all_smem.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_build_last_smem.restype = None
        L.orc_build_last_smem.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self.all_tab = np.zeros((4 ** all_bp, 32), dtype=np.uint32)
        self.last_tab = np.zeros((4 ** last_bp, 4), dtype=np.uint32)
        L.orc_build_all_smem(C.byref(self.f), all_bp, _p(self.all_tab))
        L.orc_build_last_smem(C.byref(self.f), last_bp, _p(self.last_tab))
        self.f.all_smem = self.all_tab.ctypes.data
        self.f.last_smem = self.last_tab.ctypes.data
        self.f.all_bp, self.f.last_bp = all_bp, last_bp
        return self.all_tab, self.last_tab

    def drop_fma(self):
        self.f.all_smem = None
        self.f.last_smem = None

    def occ(self, pos: int, c: int) -> int:
        return lib().orc_fmi_occ(C.byref(self.f), pos, c)

    def backward_ext(self, k, l, s, a):
        i = Smem(0, 0, 0, 0, k, l, s)
        o = Smem()
        lib().orc_backward_ext(C.byref(self.f), C.byref(i), a, C.byref(o), None)
        return o.k, o.l, o.s

    def collect_smem(self, enc, cum, opt: SeedOpt | None = None, skip=None, counters: Counters | None = None):
        opt = opt or default_seed_opt()
        nseq = len(cum) - 1
        cap = 3 * int(cum[-1] - cum[0]) + 64
        out = np.zeros(cap, dtype=SMEM_DTYPE)
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        cum = np.ascontiguousarray(cum, dtype=np.int64)
        sk = np.ascontiguousarray(skip, dtype=np.uint8) if skip is not None else None
        n = lib().orc_collect_smem(C.byref(self.f), C.byref(opt), _p(enc), _p(cum), _p(sk), nseq,
                                   _p(out), cap, C.byref(counters) if counters is not None else None)
        assert n >= 0
        return out[:n].copy()

    def sa_entry(self, pos: int) -> int:
        return lib().orc_sa_entry(C.byref(self.f), pos, None)

    def sa_lookup(self, smems, max_occ: int = 500, counters: Counters | None = None):
        n = len(smems)
        sm = np.ascontiguousarray(smems)
        cap = int(np.minimum(sm["s"], max_occ).sum()) + 1
        coord = np.zeros(cap, dtype=np.int64)
        off = np.zeros(n + 1, dtype=np.int64)
        t = lib().orc_sa_lookup(C.byref(self.f), _p(sm), n, max_occ, _p(coord), cap, _p(off),
                                C.byref(counters) if counters is not None else None)
        assert t >= 0
        return coord[:t].copy(), off


class OrcErt(C.Structure):
    _fields_ = [("kmer", C.c_int32), ("xmer", C.c_int32), ("read_len", C.c_int32), ("hit_threshold", C.c_int32),
                ("kmer_table", C.c_void_p), ("mlt", C.c_void_p), ("mlt_bytes", C.c_int64),
                ("ref", C.c_void_p), ("ref_len", C.c_int64)]


class OracleERT:
    """ERT index built by the restated writer (ert_oracle.c) over an OracleFMI; kmer_table / mlt are numpy arrays in
    the reference's file layout (<prefix>.kmer_table, <prefix>.mlt_table)."""

    def __init__(self, fmi: "OracleFMI", ref_0123, kmer: int = 15, xmer: int = 4, read_len: int = 151, hit_threshold: int = 256):
        L = lib()
        L.orc_ert_build.restype = C.c_void_p
        L.orc_ert_build.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_ert_free.restype = None
        L.orc_ert_free.argtypes = [C.c_void_p]
        L.orc_ert_profile.restype = None
        L.orc_ert_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_ert_hits.restype = C.c_int64
        L.orc_ert_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64]
        L.orc_ert_collect.restype = C.c_int64
        L.orc_ert_collect.argtypes = [C.c_void_p] * 5 + [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
        self.fmi = fmi
        self.kmer_table = np.zeros(4 ** kmer, dtype=np.uint64)
        nb = C.c_int64(0)
        ptr = L.orc_ert_build(C.byref(fmi.f), kmer, xmer, read_len, hit_threshold, _p(self.kmer_table), C.byref(nb))
        self.mlt = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(max(int(nb.value), 1),)).copy()[:int(nb.value)]
        L.orc_ert_free(ptr)
        self.mlt_pad = np.concatenate([self.mlt, np.zeros(16, dtype=np.uint8)])     # 5-/8-byte reads near the end
        self.ref = np.ascontiguousarray(ref_0123, dtype=np.uint8)
        self.e = OrcErt(kmer, xmer, read_len, hit_threshold, self.kmer_table.ctypes.data, self.mlt_pad.ctypes.data,
                        int(nb.value), self.ref.ctypes.data, len(self.ref))

    @classmethod
    def from_tables(cls, kmer_table, mlt, ref_0123, kmer: int = 15, xmer: int = 4, read_len: int = 151, hit_threshold: int = 256):
        """An index that already exists (read from files, or fetched from the GPU builder).  `mlt` must be a view of a
        buffer at least 16 bytes longer than itself (capi.Ert.fetch(pad=16)): the decoder loads 5- / 8-byte fields."""
        L = lib()
        L.orc_ert_collect.restype = C.c_int64
        L.orc_ert_collect.argtypes = [C.c_void_p] * 5 + [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
        self = cls.__new__(cls)
        self.fmi = None
        self.kmer_table = np.ascontiguousarray(kmer_table, dtype=np.uint64)
        self.mlt = mlt
        self.mlt_pad = mlt
        assert mlt.flags["C_CONTIGUOUS"] and mlt.base is not None and mlt.base.nbytes >= mlt.nbytes + 16
        self.ref = np.ascontiguousarray(ref_0123, dtype=np.uint8)
        self.e = OrcErt(kmer, xmer, read_len, hit_threshold, self.kmer_table.ctypes.data, self.mlt.ctypes.data,
                        int(len(mlt)), self.ref.ctypes.data, len(self.ref))
        return self

    def profile(self, read, i: int, M: int = 20):
        q = np.ascontiguousarray(read, dtype=np.uint8)
        out = np.zeros(M, dtype=np.uint8)
        lib().orc_ert_profile(C.byref(self.e), _p(q), len(q), i, M, _p(out))
        return out

    def hits(self, read, i: int, mlen: int, cap: int = 1 << 16):
        q = np.ascontiguousarray(read, dtype=np.uint8)
        out = np.zeros(cap, dtype=np.int64)
        n = lib().orc_ert_hits(C.byref(self.e), _p(q), len(q), i, mlen, _p(out), cap)
        return out[:min(n, cap)].copy(), n

    def collect(self, enc, cum, opt: SeedOpt | None = None, skip=None):
        """-> (smems in (rid, m, n) order with k = l = 0, sa_coord, sa_off): what collect_smem + sa_lookup give"""
        opt = opt or default_seed_opt()
        nseq = len(cum) - 1
        cap = 3 * int(cum[-1] - cum[0]) + 64
        out = np.zeros(cap, dtype=SMEM_DTYPE)
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        cum = np.ascontiguousarray(cum, dtype=np.int64)
        sk = np.ascontiguousarray(skip, dtype=np.uint8) if skip is not None else None
        sa_cap = cap * 8 + 1024
        while True:
            coord = np.zeros(sa_cap, dtype=np.int64)
            off = np.zeros(cap + 1, dtype=np.int64)
            n = lib().orc_ert_collect(C.byref(self.e), C.byref(opt), _p(enc), _p(cum), _p(sk), nseq, _p(out), cap,
                                      _p(coord), sa_cap, _p(off))
            if n == -1 and sa_cap < (1 << 30):
                sa_cap *= 8
                continue
            break
        assert n >= 0, f"orc_ert_collect -> {n}"
        return out[:n].copy(), coord[:off[n]].copy(), off[:n + 1].copy()

    def walk(self, enc, cum, opt: SeedOpt | None = None, skip=None):
        """The reference's own walk (ert_walk_oracle.c) -> (mems as the walk pushed them, mem_off, hits, hit_off, flags)."""
        opt = opt or default_seed_opt()
        nseq = len(cum) - 1
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        cum = np.ascontiguousarray(cum, dtype=np.int64)
        sk = np.ascontiguousarray(skip, dtype=np.uint8) if skip is not None else None
        mem_cap = 3 * int(cum[-1] - cum[0]) + 64
        hit_cap = mem_cap * 8 + 1024
        L = lib()
        L.orc_ert_walk.restype = C.c_int64
        L.orc_ert_walk.argtypes = [C.c_void_p] * 5 + [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                                      C.c_void_p, C.c_void_p]
        while True:
            mems = np.zeros(mem_cap, dtype=ERT_MEM_DTYPE)
            hits = np.zeros(hit_cap, dtype=np.uint64)
            mem_off = np.zeros(nseq + 1, dtype=np.int64)
            hit_off = np.zeros(nseq + 1, dtype=np.int64)
            flags = C.c_int32(0)
            n = L.orc_ert_walk(C.byref(self.e), C.byref(opt), _p(enc), _p(cum), _p(sk), nseq, _p(mems), mem_cap, _p(mem_off),
                               _p(hits), hit_cap, _p(hit_off), C.byref(flags))
            if n == -1 and hit_cap < (1 << 31):
                hit_cap *= 8
                continue
            break
        assert n >= 0, f"orc_ert_walk -> {n}"
        return mems[:n].copy(), mem_off, hits[:hit_off[nseq]].copy(), hit_off, int(flags.value)

    def walk_collect(self, enc, cum, opt: SeedOpt | None = None, skip=None):
        """The reference's own walk in collect()'s layout -> (smems, sa_coord, sa_off, cls, flags); cls bit 0 forward,
        bit 1 fetch_leaves, bit 2 end_correction != 0."""
        opt = opt or default_seed_opt()
        nseq = len(cum) - 1
        cap = 3 * int(cum[-1] - cum[0]) + 64
        out = np.zeros(cap, dtype=SMEM_DTYPE)
        cls = np.zeros(cap, dtype=np.uint8)
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        cum = np.ascontiguousarray(cum, dtype=np.int64)
        sk = np.ascontiguousarray(skip, dtype=np.uint8) if skip is not None else None
        sa_cap = cap * 8 + 1024
        L = lib()
        L.orc_ert_walk_collect.restype = C.c_int64
        L.orc_ert_walk_collect.argtypes = [C.c_void_p] * 5 + [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                                              C.c_void_p, C.c_void_p]
        while True:
            coord = np.zeros(sa_cap, dtype=np.int64)
            off = np.zeros(cap + 1, dtype=np.int64)
            flags = C.c_int32(0)
            n = L.orc_ert_walk_collect(C.byref(self.e), C.byref(opt), _p(enc), _p(cum), _p(sk), nseq, _p(out), cap, _p(coord),
                                       sa_cap, _p(off), _p(cls), C.byref(flags))
            if n == -1 and sa_cap < (1 << 30):
                sa_cap *= 8
                continue
            break
        assert n >= 0, f"orc_ert_walk_collect -> {n}"
        return out[:n].copy(), coord[:off[n]].copy(), off[:n + 1].copy(), cls[:n].copy(), int(flags.value)


def bsw_pairs(pairs, ref, qer, w: int, opt: SwOpt | None = None):
    """Run the restated scalar extension over a SeqPair array (returns a copy, cells)."""
    opt = opt or default_sw_opt()
    p = np.ascontiguousarray(pairs).copy()
    cells = C.c_int64(0)
    lib().orc_bsw_pairs(C.byref(opt), _p(p), _p(np.ascontiguousarray(ref)), _p(np.ascontiguousarray(qer)),
                        len(p), w, C.byref(cells))
    return p, cells.value


# ---------------------------------------------------------------------------
# the real reference objects (oracle/_ref), when present
# ---------------------------------------------------------------------------
def _cpu_flags():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


def ref_lib(isa: str | None = None):
    """CDLL of oracle/_ref/libref_sw_<isa>.so or None.  isa None = widest the host runs."""
    flags = _cpu_flags()
    order = [isa] if isa else [i for i, fl in (("avx512", "avx512bw"), ("avx2", "avx2"), ("sse41", "sse4_1"))
                               if fl in flags]
    for i in order:
        path = os.path.join(HERE, "_ref", f"libref_sw_{i}.so")
        if os.path.exists(path):
            L = C.CDLL(path)
            vp = C.c_void_p
            for fn in ("ref_bsw_scalar", "ref_bsw_vec16", "ref_bsw_vec8"):
                getattr(L, fn).restype = None
                getattr(L, fn).argtypes = [vp, vp, vp, vp, C.c_int, C.c_int]
            L.ref_ksw_global2.restype = C.c_int
            L.ref_ksw_global2.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int]
            L.ref_ksw_extend2.restype = C.c_int
            L.ref_ksw_extend2.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int] + [vp] * 5
            L.isa = i
            return L
    return None


def ref_bsw(L, which: str, pairs, ref, qer, w: int, opt: SwOpt | None = None):
    """Run the reference's scalar / 16-bit / 8-bit BSW over a SeqPair array (copy returned)."""
    opt = opt or default_sw_opt()
    n = len(pairs)
    width = 64
    p = np.zeros(n + width, dtype=SEQPAIR_DTYPE)       # room for SIMD-width padding entries
    p[:n] = pairs
    r = np.concatenate([np.ascontiguousarray(ref), np.zeros(4096, np.uint8)])
    q = np.concatenate([np.ascontiguousarray(qer), np.zeros(4096, np.uint8)])
    getattr(L, {"scalar": "ref_bsw_scalar", "vec16": "ref_bsw_vec16", "vec8": "ref_bsw_vec8"}[which])(
        C.byref(opt), _p(p), _p(r), _p(q), n, w)
    return p[:n].copy()


# ---------------------------------------------------------------------------
# chaining / chain filter / chain -> alignment regions (chain_oracle.c)
# ---------------------------------------------------------------------------
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
assert CONTIG_DTYPE.itemsize == 16 and CHAIN_SEED_DTYPE.itemsize == 32 and CHAIN_DTYPE.itemsize == 48
assert ALNREG_DTYPE.itemsize == 112


class MemOpt(C.Structure):
    _fields_ = [("a", C.c_int32), ("o_del", C.c_int32), ("e_del", C.c_int32), ("o_ins", C.c_int32), ("e_ins", C.c_int32),
                ("pen_clip5", C.c_int32), ("pen_clip3", C.c_int32), ("w", C.c_int32), ("zdrop", C.c_int32),
                ("min_seed_len", C.c_int32), ("min_chain_weight", C.c_int32), ("max_chain_extend", C.c_int32),
                ("max_occ", C.c_int32), ("max_chain_gap", C.c_int32), ("mask_level", C.c_float),
                ("drop_ratio", C.c_float), ("mat", C.c_int8 * 25), ("pad_", C.c_int8 * 3), ("extend_all", C.c_int32), ("mask_level_redun", C.c_float), ("max_ins", C.c_int32),
                ("b", C.c_int32), ("pen_unpaired", C.c_int32), ("max_matesw", C.c_int32), ("mapq_coef_len", C.c_int32)]


def default_mem_opt(a: int = 1, b: int = 4) -> MemOpt:
    """mem_opt_init defaults (src/bwamem.cpp:135-171)."""
    o = MemOpt(a, 6, 1, 6, 1, 5, 5, 100, 100, 19, 0, 1 << 30, 500, 10000, 0.5, 0.5)
    o.mask_level_redun = 0.95
    o.max_ins = 10000
    o.b, o.pen_unpaired, o.max_matesw = b, 17, 50
    o.mapq_coef_len = 50
    for i, v in enumerate(fill_scmat(a, b)):
        o.mat[i] = v
    return o


class OrcBns(C.Structure):
    _fields_ = [("l_pac", C.c_int64), ("n_seqs", C.c_int32), ("contigs", C.c_void_p)]


class TaskDump(C.Structure):
    _fields_ = [("build_only", C.c_int32), ("pad_", C.c_int32), ("n_left", C.c_int64), ("n_right", C.c_int64),
                ("left", C.c_void_p), ("right", C.c_void_p), ("left_ref", C.c_void_p), ("left_qer", C.c_void_p),
                ("right_ref", C.c_void_p), ("right_qer", C.c_void_p), ("left_ref_bytes", C.c_int64),
                ("left_qer_bytes", C.c_int64), ("right_ref_bytes", C.c_int64), ("right_qer_bytes", C.c_int64)]


def single_contig(l_pac: int):
    c = np.zeros(1, CONTIG_DTYPE)
    c["len"] = l_pac
    return c


def _bns(l_pac, contigs):
    contigs = np.ascontiguousarray(contigs, dtype=CONTIG_DTYPE)
    return OrcBns(int(l_pac), len(contigs), contigs.ctypes.data), contigs


def kbt_script(pos, do_put, L=None):
    """(lower ids, traversal order) of the restated B-tree — or of the reference's kbtree.h when L is ref_chain_lib()."""
    pos = np.ascontiguousarray(pos, np.int64)
    do_put = np.ascontiguousarray(do_put, np.uint8)
    lower = np.zeros(len(pos), np.int32)
    order = np.zeros(len(pos), np.int32)
    fn = L.ref_kbt_script if L is not None else lib().orc_kbt_script
    m = fn(len(pos), _p(pos), _p(do_put), _p(lower), _p(order))
    return lower, order[:m].copy()


def flt_sort(w, L=None):
    w = np.ascontiguousarray(w, np.uint32)
    order = np.zeros(len(w), np.int32)
    fn = L.ref_flt_sort if L is not None else lib().orc_flt_sort
    fn(len(w), _p(w), _p(order))
    return order


def ref_chain_lib():
    path = os.path.join(HERE, "_ref", "libref_chain.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.ref_kbt_script.restype = C.c_int64
    L.ref_kbt_script.argtypes = [C.c_int64] + [C.c_void_p] * 4
    L.ref_flt_sort.restype = None
    L.ref_flt_sort.argtypes = [C.c_int64, C.c_void_p, C.c_void_p]
    L.ref_ars_sort.restype = None
    L.ref_ars_sort.argtypes = [C.c_int64, C.c_int] + [C.c_void_p] * 4
    return L


def chain_seeds(smems, sa_coord, sa_off, cum, l_pac, contigs=None, opt: MemOpt | None = None, do_flt: bool | int = True,
                ref_string=None, enc=None):
    """Restated mem_chain_seeds (+ mem_chain_flt and mem_flt_chained_seeds when do_flt) -> (chains, seeds, chain_off).
    ref_string / enc are only needed when a read is long enough (~1100 bases) for mem_flt_chained_seeds to act."""
    opt = opt or default_mem_opt()
    bns, keep = _bns(l_pac, contigs if contigs is not None else single_contig(l_pac))
    smems = np.ascontiguousarray(smems, dtype=SMEM_DTYPE)
    sa_coord = np.ascontiguousarray(sa_coord, np.int64)
    sa_off = np.ascontiguousarray(sa_off, np.int64)
    cum = np.ascontiguousarray(cum, np.int64)
    nseq = len(cum) - 1
    cap = max(1, len(sa_coord))
    chains = np.zeros(cap, CHAIN_DTYPE)
    seeds = np.zeros(cap, CHAIN_SEED_DTYPE)
    chain_off = np.zeros(nseq + 1, np.int64)
    n_seeds = C.c_int64(0)
    n = lib().orc_chain_seeds(C.byref(opt), C.byref(bns), _p(smems), len(smems), _p(sa_coord), _p(sa_off), _p(cum), nseq,
                              int(do_flt), _p(chains), cap, _p(seeds), cap, _p(chain_off), C.byref(n_seeds),
                              _p(np.ascontiguousarray(ref_string, np.uint8)) if ref_string is not None else None,
                              _p(np.ascontiguousarray(enc, np.uint8)) if enc is not None else None)
    assert n >= 0, n
    # compact the seed array to the kept chains, in chain order
    chains = chains[:n].copy()
    out = np.zeros(int(chains["n"].sum()), CHAIN_SEED_DTYPE)
    o = 0
    for c in chains:
        k = int(c["n"])
        out[o:o + k] = seeds[c["seed_off"]:c["seed_off"] + k]
        c["seed_off"] = o
        o += k
    return chains, out, chain_off


def chain2aln(chains, seeds, chain_off, enc, cum, ref_string, l_pac, contigs=None, opt: MemOpt | None = None,
              build_only: bool = False, want_tasks: bool = False):
    """Restated mem_chain2aln_across_reads_V2 -> (regs, reg_off, seeds with .aln[, tasks])."""
    opt = opt or default_mem_opt()
    bns, keep = _bns(l_pac, contigs if contigs is not None else single_contig(l_pac))
    chains = np.ascontiguousarray(chains, dtype=CHAIN_DTYPE)
    seeds = np.ascontiguousarray(seeds, dtype=CHAIN_SEED_DTYPE).copy()
    chain_off = np.ascontiguousarray(chain_off, np.int64)
    cum = np.ascontiguousarray(cum, np.int64)
    enc = np.ascontiguousarray(enc, np.uint8)
    ref_string = np.ascontiguousarray(ref_string, np.uint8)
    nseq = len(cum) - 1
    cap = max(1, len(seeds))
    regs = np.zeros(cap, ALNREG_DTYPE)
    reg_off = np.zeros(nseq + 1, np.int64)
    dump = TaskDump()
    dump.build_only = int(build_only)
    use_dump = build_only or want_tasks
    n = lib().orc_chain2aln(C.byref(opt), C.byref(bns), _p(ref_string), _p(enc), _p(cum), nseq, _p(chains), _p(chain_off),
                            _p(seeds), _p(regs), cap, _p(reg_off), C.byref(dump) if use_dump else None)
    assert n >= 0, n
    res = [regs[:n].copy(), reg_off, seeds]
    if use_dump:
        def arr(ptr, count, dt):
            if not count:
                return np.zeros(0, dt)
            buf = (C.c_uint8 * (count * np.dtype(dt).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dt).copy()
        res.append({
            "left": arr(dump.left, dump.n_left, SEQPAIR_DTYPE), "right": arr(dump.right, dump.n_right, SEQPAIR_DTYPE),
            "left_ref": arr(dump.left_ref, dump.left_ref_bytes, np.uint8), "left_qer": arr(dump.left_qer, dump.left_qer_bytes, np.uint8),
            "right_ref": arr(dump.right_ref, dump.right_ref_bytes, np.uint8), "right_qer": arr(dump.right_qer, dump.right_qer_bytes, np.uint8)})
        lib().orc_task_dump_free(C.byref(dump))
    return tuple(res)


# ---------------------------------------------------------------------------
# the tail of mem_kernel2_core (dedup_oracle.c)
# ---------------------------------------------------------------------------
def ksw_global2_score(query, target, w: int, opt: SwOpt | None = None, L=None):
    """Score of the banded global alignment (ksw_global2 without backtrack); L = ref_lib() runs the reference's."""
    opt = opt or default_sw_opt()
    q = np.ascontiguousarray(query, np.uint8)
    t = np.ascontiguousarray(target, np.uint8)
    if L is not None:
        return L.ref_ksw_global2(C.byref(opt), len(q), _p(q), len(t), _p(t), w)
    return lib().orc_ksw_global2_score(len(q), _p(q), len(t), _p(t), C.cast(opt.mat, C.c_void_p), opt.o_del, opt.e_del,
                                       opt.o_ins, opt.e_ins, w)


ALN_DTYPE = np.dtype([("pos", "<i8"), ("rid", "<i4"), ("flag", "<i4"), ("is_rev", "<i4"), ("is_alt", "<i4"), ("mapq", "<i4"),
                      ("NM", "<i4"), ("n_cigar", "<i4"), ("md_len", "<i4"), ("cigar_off", "<i8"), ("md_off", "<i8"),
                      ("score", "<i4"), ("sub", "<i4"), ("alt_sc", "<i4"), ("pad_", "<i4")])
assert ALN_DTYPE.itemsize == 72


def ksw_global2_cigar(query, target, w: int, opt: SwOpt | None = None, L=None):
    """(score, cigar uint32[]) of ksw_global2 with traceback; L = a reference library from ref_lib() to run the real one."""
    opt = opt or default_sw_opt()
    q = np.ascontiguousarray(query, np.uint8)
    t = np.ascontiguousarray(target, np.uint8)
    cig = np.zeros(len(q) + len(t) + 4, np.uint32)
    n = C.c_int(0)
    if L is None:
        f = lib().orc_ksw_global2_cigar
        f.restype = C.c_int
        sc = f(len(q), _p(q), len(t), _p(t), C.byref(opt, SwOpt.mat.offset), opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, w,
               C.byref(n), _p(cig))
    else:
        L.ref_ksw_global2_cigar.restype = C.c_int
        L.ref_ksw_global2_cigar.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        sc = L.ref_ksw_global2_cigar(C.byref(opt), len(q), _p(q), len(t), _p(t), w, C.byref(n), _p(cig), len(cig))
    return sc, cig[:n.value].copy()


def reg2aln(regs, reg_off, enc, cum, ref_string, l_pac, contigs=None, opt: MemOpt | None = None):
    """mem_reg2aln over every region (regions grouped by read): (aln records, cigar pool, md pool as bytes)."""
    opt = opt or default_mem_opt()
    bns, keep = _bns(l_pac, contigs if contigs is not None else single_contig(l_pac))
    regs = np.ascontiguousarray(regs, dtype=ALNREG_DTYPE)
    enc = np.ascontiguousarray(enc, np.uint8)
    ref_string = np.ascontiguousarray(ref_string, np.uint8)
    out = np.zeros(len(regs), ALN_DTYPE)
    cigs, mds = [], []
    co = mo = 0
    f = lib().orc_reg2aln
    f.restype = C.c_int
    for r in range(len(reg_off) - 1):
        q = enc[cum[r]:cum[r + 1]]
        for k in range(int(reg_off[r]), int(reg_off[r + 1])):
            ar = regs[k:k + 1]
            span = max(int(ar["re"][0] - ar["rb"][0]), 0)
            cig = np.zeros(len(q) + span + 8, np.uint32)
            md = np.zeros(3 * span + 32, np.uint8)
            f(C.byref(opt), C.byref(bns), _p(ref_string), len(q), _p(q), _p(ar), _p(out[k:k + 1]), _p(cig), _p(md))
            out[k]["cigar_off"], out[k]["md_off"] = co, mo
            cigs.append(cig[:out[k]["n_cigar"]]); mds.append(md[:out[k]["md_len"]])
            co += int(out[k]["n_cigar"]); mo += int(out[k]["md_len"])
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)      # noqa: E731
    return out, cat(cigs, np.uint32), cat(mds, np.uint8)


def fastq_parse(text: bytes):
    """kseq_read over a buffer -> dict(n, names[list of bytes], comments[list of bytes or None], enc, cum, quals, has_qual, status).
    status 0: the whole buffer was read; -2: the reader stopped at a quality string of the wrong length."""
    n = len(text)
    cap = n + 16
    max_reads = text.count(b"@") + text.count(b">") + 1
    names = C.create_string_buffer(cap); comments = C.create_string_buffer(cap); qual = C.create_string_buffer(cap)
    seq = np.zeros(cap, np.uint8)
    noff = np.zeros(max_reads + 1, np.int64); coff = np.zeros(max_reads + 1, np.int64); cum = np.zeros(max_reads + 1, np.int64)
    hq = np.zeros(max_reads + 1, np.uint8)
    f = lib().orc_fastq_parse
    f.restype = C.c_int64
    nr = f(text, C.c_int64(n), C.c_int64(max_reads), names, _p(noff), comments, _p(coff), _p(seq), qual, _p(cum), _p(hq))
    status = 0
    if nr < 0:
        status, nr = -2, -2 - nr
    nm = [names.raw[noff[i]:noff[i + 1]] for i in range(nr)]
    cm = [comments.raw[coff[i]:coff[i + 1]] or None for i in range(nr)]
    return dict(n=nr, names=nm, comments=cm, enc=seq[:cum[nr]].copy(), cum=cum[:nr + 1].copy(),
                quals=np.frombuffer(qual.raw[:cum[nr]], np.uint8).copy(), has_qual=hq[:nr].copy(), status=status)


class SamOpt(C.Structure):
    """bwams_sam_opt_t (include/bwams_types.h)."""
    _fields_ = [("T", C.c_int32), ("flag", C.c_int32), ("XA_drop_ratio", C.c_float), ("max_XA_hits", C.c_int32),
                ("max_XA_hits_alt", C.c_int32), ("rg_id", C.c_char * 256)]


def default_sam_opt(flag: int = 0, rg_id: bytes = b"") -> SamOpt:
    """mem_opt_init defaults (src/bwamem.cpp:135-171)."""
    return SamOpt(30, flag, 0.80, 5, 200, rg_id)


def contig_name_table(names):
    """(NUL-terminated names back to back, int32 start offsets) as the SAM restatement and the C-ABI take them."""
    blob, off = bytearray(), []
    for nm in names:
        off.append(len(blob))
        blob += (nm if isinstance(nm, bytes) else nm.encode()) + b"\0"
    off.append(len(blob))
    return bytes(blob), np.asarray(off, np.int32)


@contextlib.contextmanager
def contig_annos(annos):
    """Within the block the SAM restatement knows the sequences' annotations (bntann1_t.anno, b"" = none): MEM_F_REF_HDR's XR tags."""
    blob, off = contig_name_table(annos)
    f = lib().orc_set_contig_annos
    f.restype = None
    f(blob, _p(off))
    try:
        yield
    finally:
        f(None, None)


def reg2sam_se(regs, reg_off, enc, cum, ref_string, l_pac, names, quals=None, comments=None, contigs=None, contig_names=None,
               opt: MemOpt | None = None, sopt: SamOpt | None = None):
    """mem_reg2sam (single-end) over every read of a chunk: list of bytes, one SAM text block per read.
    names / comments: lists of bytes; quals: uint8 array laid out like enc (cum offsets) or None."""
    opt = opt or default_mem_opt()
    sopt = sopt or default_sam_opt()
    contigs = contigs if contigs is not None else single_contig(l_pac)
    bns, keep = _bns(l_pac, contigs)
    cn_blob, cn_off = contig_name_table(contig_names if contig_names is not None else [b"chr%d" % (i + 1) for i in range(len(keep))])
    regs = np.ascontiguousarray(regs, dtype=ALNREG_DTYPE)
    enc = np.ascontiguousarray(enc, np.uint8)
    ref_string = np.ascontiguousarray(ref_string, np.uint8)
    f = lib().orc_reg2sam_se
    f.restype = C.c_int64
    out = []
    cap = 1 << 16
    buf = C.create_string_buffer(cap)
    for r in range(len(reg_off) - 1):
        q = enc[cum[r]:cum[r + 1]]
        qual = (bytes(quals[cum[r]:cum[r + 1]]) or None) if quals is not None else None      # kseq2bseq1: empty = none
        a = regs[int(reg_off[r]):int(reg_off[r + 1])]
        cm = comments[r] if comments is not None else None
        while True:
            n = f(C.byref(opt), C.byref(sopt), C.byref(bns), cn_blob, _p(cn_off), _p(ref_string), len(q), _p(q), qual, names[r], cm,
                  _p(a) if len(a) else None, len(a), buf, cap)
            if n >= 0:
                break
            cap = max(2 * cap, -n + 16)
            buf = C.create_string_buffer(cap)
        out.append(buf.raw[:n])
    return out


def perfect2sam(regs, read, l_pac, seed_len, name, qual=None, comment=None, contigs=None, contig_names=None, opt: MemOpt | None = None,
                sopt: SamOpt | None = None):
    """mem_perfect2sam_cont for one resolved read from its mem_perfect2reg regions -> bytes."""
    opt = opt or default_mem_opt()
    sopt = sopt or default_sam_opt()
    contigs = contigs if contigs is not None else single_contig(l_pac)
    bns, keep = _bns(l_pac, contigs)
    cn_blob, cn_off = contig_name_table(contig_names if contig_names is not None else [b"chr%d" % (i + 1) for i in range(len(keep))])
    regs = np.ascontiguousarray(regs, dtype=ALNREG_DTYPE)
    q = np.ascontiguousarray(read, np.uint8)
    f = lib().orc_perfect2sam
    f.restype = C.c_int64
    cap = 4096 + 1024 * len(regs)
    buf = C.create_string_buffer(cap)
    n = f(C.byref(opt), C.byref(sopt), C.byref(bns), cn_blob, _p(cn_off), int(seed_len), len(q), _p(q), qual, name, comment, _p(regs), len(regs),
          buf, cap)
    assert n >= 0
    return buf.raw[:n]


def sam_pe(regs, reg_off, enc, cum, ref_string, l_pac, pes, pairs, names, quals=None, comments=None, contigs=None, contig_names=None,
           opt: MemOpt | None = None, sopt: SamOpt | None = None):
    """mem_sam_pe from mem_pair's result on, over every pair of a chunk (regs / reg_off / pairs = pair_pe's outputs):
    list of bytes, one SAM text block per read (2p, 2p + 1 = the ends of pair p)."""
    opt = opt or default_mem_opt()
    sopt = sopt or default_sam_opt()
    contigs = contigs if contigs is not None else single_contig(l_pac)
    bns, keep = _bns(l_pac, contigs)
    cn_blob, cn_off = contig_name_table(contig_names if contig_names is not None else [b"chr%d" % (i + 1) for i in range(len(keep))])
    regs = np.ascontiguousarray(regs, dtype=ALNREG_DTYPE).copy()          # edited in place, as the reference does
    enc = np.ascontiguousarray(enc, np.uint8)
    ref_string = np.ascontiguousarray(ref_string, np.uint8)
    pes = np.ascontiguousarray(pes, PESTAT_DTYPE)
    pairs = np.ascontiguousarray(pairs, PAIR_DTYPE)
    f = lib().orc_sam_pe
    f.restype = None
    out = []
    cap = 1 << 16
    bufs = [C.create_string_buffer(cap), C.create_string_buffer(cap)]
    for p in range(len(pairs)):
        r0 = 2 * p
        lens = (C.c_int32 * 2)(int(cum[r0 + 1] - cum[r0]), int(cum[r0 + 2] - cum[r0 + 1]))
        seqs = (C.c_void_p * 2)(enc.ctypes.data + int(cum[r0]), enc.ctypes.data + int(cum[r0 + 1]))
        qb = [(bytes(quals[cum[r0 + i]:cum[r0 + i + 1]]) or None) if quals is not None else None for i in range(2)]
        qs = (C.c_char_p * 2)(qb[0], qb[1])
        nm = (C.c_char_p * 2)(names[r0], names[r0 + 1])
        cm = (C.c_char_p * 2)(comments[r0] if comments is not None else None, comments[r0 + 1] if comments is not None else None)
        nreg = (C.c_int32 * 2)(int(reg_off[r0 + 1] - reg_off[r0]), int(reg_off[r0 + 2] - reg_off[r0 + 1]))
        rp = (C.c_void_p * 2)(regs.ctypes.data + int(reg_off[r0]) * ALNREG_DTYPE.itemsize, regs.ctypes.data + int(reg_off[r0 + 1]) * ALNREG_DTYPE.itemsize)
        while True:
            keep_regs = regs[int(reg_off[r0]):int(reg_off[r0 + 2])].copy()
            ob = (C.c_void_p * 2)(C.addressof(bufs[0]), C.addressof(bufs[1]))
            caps = (C.c_int64 * 2)(cap, cap)
            ln = (C.c_int64 * 2)(0, 0)
            f(C.byref(opt), C.byref(sopt), C.byref(bns), cn_blob, _p(cn_off), _p(ref_string), _p(pes), lens, seqs, qs, nm, cm, rp, nreg,
              pairs[p:p + 1].ctypes.data_as(C.c_void_p), ob, caps, ln)
            if ln[0] >= 0 and ln[1] >= 0:
                break
            regs[int(reg_off[r0]):int(reg_off[r0 + 2])] = keep_regs          # undo the edits before the retry
            cap = max(2 * cap, -min(ln[0], ln[1]) + 16)
            bufs = [C.create_string_buffer(cap), C.create_string_buffer(cap)]
        out.append(bufs[0].raw[:ln[0]]); out.append(bufs[1].raw[:ln[1]])
    return out


def ars_sort(which: int, k0, k1=None, k2=None, L=None):
    """Order of ks_introsort(mem_ars2) (which = 0, key re) / ks_introsort(mem_ars) (which = 1, keys score, rb, qb)."""
    k0 = np.ascontiguousarray(k0, np.int64)
    k1 = np.ascontiguousarray(k1 if k1 is not None else np.zeros(len(k0)), np.int64)
    k2 = np.ascontiguousarray(k2 if k2 is not None else np.zeros(len(k0)), np.int64)
    order = np.zeros(len(k0), np.int32)
    (L.ref_ars_sort if L is not None else lib().orc_ars_sort)(len(k0), which, _p(k0), _p(k1), _p(k2), _p(order))
    return order


def regs_finish(regs, reg_off, enc, cum, ref_string, l_pac, contigs=None, opt: MemOpt | None = None):
    """Restated tail of mem_kernel2_core: drop purged regions, mem_sort_dedup_patch, ALT mark -> (regs, reg_off)."""
    opt = opt or default_mem_opt()
    bns, keep = _bns(l_pac, contigs if contigs is not None else single_contig(l_pac))
    regs = np.ascontiguousarray(regs, dtype=ALNREG_DTYPE).copy()
    reg_off = np.ascontiguousarray(reg_off, np.int64)
    cum = np.ascontiguousarray(cum, np.int64)
    enc = np.ascontiguousarray(enc, np.uint8)
    ref_string = np.ascontiguousarray(ref_string, np.uint8)
    out_off = np.zeros(len(cum), np.int64)
    n = lib().orc_regs_finish(C.byref(opt), C.byref(bns), _p(ref_string), _p(enc), _p(cum), len(cum) - 1, _p(regs), _p(reg_off),
                              _p(out_off))
    return regs[:n].copy(), out_off


PESTAT_DTYPE = np.dtype([("low", "<i4"), ("high", "<i4"), ("failed", "<i4"), ("pad_", "<i4"), ("avg", "<f8"), ("std", "<f8")])
assert PESTAT_DTYPE.itemsize == 32


def pestat(regs, reg_off, l_pac, opt: MemOpt | None = None):
    """Restated mem_pestat over final regions (reads 2i, 2i+1 = pair i) -> 4 records (FF, FR, RF, RR)."""
    opt = opt or default_mem_opt()
    regs = np.ascontiguousarray(regs, dtype=ALNREG_DTYPE)
    reg_off = np.ascontiguousarray(reg_off, np.int64)
    pes = np.zeros(4, PESTAT_DTYPE)
    lib().orc_pestat(C.byref(opt), int(l_pac), len(reg_off) - 1, _p(regs), _p(reg_off), _p(pes))
    return pes


PAIR_DTYPE = np.dtype([("score", "<i4"), ("sub", "<i4"), ("n_sub", "<i4"), ("z", "<i4", (2,)), ("n_pri", "<i4", (2,)),
                       ("n_matesw", "<i4")])
assert PAIR_DTYPE.itemsize == 32


def pair_pe(regs, reg_off, enc, cum, ref_string, l_pac, pes, contigs=None, opt: MemOpt | None = None, id_base: int = 0,
            no_rescue: bool = False, use_ert: bool = False, no_pairing: bool = False, primary5_T: int = -1):
    """Restated PE tail up to the pairing decision: mate rescue, mem_mark_primary_se, [mem_reorder_primary5,] mem_pair
    -> (regs, reg_off, pairs).  Reads 2p, 2p + 1 are the ends of pair p."""
    opt = opt or default_mem_opt()
    bns, keep = _bns(l_pac, contigs if contigs is not None else single_contig(l_pac))
    regs = np.ascontiguousarray(regs, dtype=ALNREG_DTYPE)
    reg_off = np.ascontiguousarray(reg_off, np.int64)
    cum = np.ascontiguousarray(cum, np.int64)
    enc = np.ascontiguousarray(enc, np.uint8)
    ref_string = np.ascontiguousarray(ref_string, np.uint8)
    pes = np.ascontiguousarray(pes, PESTAT_DTYPE)
    nseq = len(cum) - 1
    assert nseq % 2 == 0 and len(reg_off) == nseq + 1
    cap = len(regs) + 4 * min(opt.max_matesw, max(1, len(regs))) * nseq + 16
    out = np.zeros(cap, ALNREG_DTYPE)
    out_off = np.zeros(nseq + 1, np.int64)
    pairs = np.zeros(nseq // 2, PAIR_DTYPE)
    L = lib()
    L.orc_pair_pe.restype = C.c_int64
    n = L.orc_pair_pe(C.byref(opt), C.byref(bns), _p(ref_string), _p(enc), _p(cum), nseq // 2, _p(regs), _p(reg_off), _p(pes),
                      C.c_int64(id_base), int(no_rescue) | (int(use_ert) << 1) | (int(no_pairing) << 2), int(primary5_T), _p(out),
                      C.c_int64(cap), _p(out_off), _p(pairs))
    assert n >= 0, "orc_pair_pe: output capacity"
    return out[:n].copy(), out_off, pairs


def mark_primary_se(regs, id_: int, opt: MemOpt | None = None, primary5_T: int = -1):
    """Restated mem_mark_primary_se [+ mem_reorder_primary5(T)] on one read's regions -> (regs, n_pri)."""
    opt = opt or default_mem_opt()
    regs = np.ascontiguousarray(regs, dtype=ALNREG_DTYPE).copy()
    n_pri = lib().orc_mark_primary_se(C.byref(opt), len(regs), _p(regs), C.c_int64(id_))
    if primary5_T >= 0:
        regs = reorder_primary5(regs, primary5_T)
    return regs, n_pri


def reorder_primary5(regs, T: int):
    """Restated mem_reorder_primary5 (bwamem.cpp:2009-2031) on one read's marked regions -> regs."""
    regs = np.ascontiguousarray(regs, dtype=ALNREG_DTYPE).copy()
    lib().orc_reorder_primary5.restype = None
    lib().orc_reorder_primary5(int(T), len(regs), _p(regs))
    return regs


def pestat_keys(regs, reg_off, l_pac, opt: MemOpt | None = None):
    """The insert-size keys (orientation << 60 | insert size) of the qualifying pairs, in pair order."""
    opt = opt or default_mem_opt()
    regs = np.ascontiguousarray(regs, dtype=ALNREG_DTYPE)
    reg_off = np.ascontiguousarray(reg_off, np.int64)
    keys = np.zeros(max((len(reg_off) - 1) // 2, 1), np.uint64)
    L = lib()
    L.orc_pestat_keys.restype = C.c_int64
    n = L.orc_pestat_keys(C.byref(opt), C.c_int64(int(l_pac)), len(reg_off) - 1, _p(regs), _p(reg_off), _p(keys))
    return keys[:n].copy()


ERT_MEM_DTYPE = np.dtype([("forward", "u1"), ("pad_", "u1", (3,)), ("start", "<i4"), ("end", "<i4"), ("rc_start", "<i4"),
                          ("rc_end", "<i4"), ("skip_ref_fetch", "<i4"), ("fetch_leaves", "<i4"), ("hitbeg", "<i4"),
                          ("hitcount", "<i4"), ("end_correction", "<i4"), ("is_multi_hit", "<i4"), ("c_pivot", "<i4"),
                          ("p_pivot", "<i4"), ("pp_pivot", "<i4")])
assert ERT_MEM_DTYPE.itemsize == 56


def chain_new_ert(mems, mem_off, hits, hit_off, cum, l_pac, contigs=None, opt: MemOpt | None = None, do_flt: bool = True,
                  ref_string=None, enc=None):
    """Restated tail of mem_kernel1_core_ert: introsort of the MEMs, mem_chain_new, mem_chain_flt,
    mem_flt_chained_seeds -> (chains, seeds, chain_off)."""
    opt = opt or default_mem_opt()
    bns, keep = _bns(l_pac, contigs if contigs is not None else single_contig(l_pac))
    mems = np.ascontiguousarray(mems, dtype=ERT_MEM_DTYPE)
    mem_off = np.ascontiguousarray(mem_off, np.int64)
    hits = np.ascontiguousarray(hits, np.uint64)
    hit_off = np.ascontiguousarray(hit_off, np.int64)
    cum = np.ascontiguousarray(cum, np.int64)
    nseq = len(cum) - 1
    cap = max(1, int(np.minimum(mems["hitcount"], opt.max_occ).sum()))
    chains = np.zeros(cap, CHAIN_DTYPE)
    seeds = np.zeros(cap, CHAIN_SEED_DTYPE)
    chain_off = np.zeros(nseq + 1, np.int64)
    n_seeds = C.c_int64(0)
    L = lib()
    L.orc_chain_new_ert.restype = C.c_int64
    n = L.orc_chain_new_ert(C.byref(opt), C.byref(bns), _p(mems), _p(mem_off), _p(hits), _p(hit_off), _p(cum), nseq, int(do_flt),
                            _p(chains), C.c_int64(cap), _p(seeds), C.c_int64(cap), _p(chain_off), C.byref(n_seeds),
                            _p(np.ascontiguousarray(ref_string, np.uint8)) if ref_string is not None else None,
                            _p(np.ascontiguousarray(enc, np.uint8)) if enc is not None else None)
    assert n >= 0, n
    chains = chains[:n].copy()
    out = np.zeros(int(chains["n"].sum()), CHAIN_SEED_DTYPE)
    o = 0
    for c in chains:
        k = int(c["n"])
        out[o:o + k] = seeds[c["seed_off"]:c["seed_off"] + k]
        c["seed_off"] = o
        o += k
    return chains, out, chain_off


def ref_fmi_lib():
    """CDLL of oracle/_ref/libref_fmi.so (the reference's FMI_search.h + sais.h, oracle/ref_harness_fmi.cpp) or None."""
    path = os.path.join(HERE, "_ref", "libref_fmi.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.ref_get_occ.restype = C.c_int64
    L.ref_get_occ.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    L.ref_sais.restype = C.c_int
    L.ref_sais.argtypes = [C.c_char_p, C.c_int64, C.c_void_p]
    L.ref_fmi_sizes.restype = C.c_int
    return L
