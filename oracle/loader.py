"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Builds oracle/_build/liboracle.so with gcc on first use; loads the real
reference objects from oracle/_ref/ when they exist (built by oracle/Makefile in
the container where /root/reference is mounted; they travel to the GPU box as
prebuilt files).
"""
from __future__ import annotations

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
    srcs = [os.path.join(HERE, f) for f in ("fmi_oracle.c", "bsw_oracle.c", "ksw_oracle.c", "emf_oracle.c", "bwams_oracle.h")]
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
