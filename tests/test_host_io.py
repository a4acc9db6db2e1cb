"""The file ends of the pipeline (bwa-mem-scale_amd/host/fastq_io.cpp): the gz / plain reader that cuts chunks where bseq_read_orig
cuts them, bseq1_t records from a chunk in place (step 0), the per-shard ordered writer — and a .gz file streamed through the compiled
mem_process_seqs() against bwams_process_chunk on the same chunks.  (GPU box: the chunk buffers are page-locked.)"""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

from bwams import capi, simulate
from oracle import loader

pytestmark = pytest.mark.gpu


def _reader_chunks(path, chunk_bases, paired=False, n_buffers=2, hold=False):
    L = capi.lib()
    L.bwams_reader_error.restype = C.c_char_p
    r = C.c_void_p()
    capi._chk(L.bwams_reader_open(path.encode(), C.c_int64(chunk_bases), 1 if paired else 0, C.c_int64(0), n_buffers, C.byref(r)), "bwams_reader_open")
    out = []
    while True:
        text, nb, nr, nbases = C.c_void_p(), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        rc = L.bwams_reader_next(r, C.byref(text), C.byref(nb), C.byref(nr), C.byref(nbases))
        if rc == 1:
            break
        assert rc == 0, (rc, L.bwams_reader_error(r))
        out.append((C.string_at(text.value, nb.value), nr.value, nbases.value))
        capi._chk(L.bwams_reader_release(r, text), "bwams_reader_release")
    L.bwams_reader_close(r)
    return out


def _expected_cuts(text, chunk_bases, paired):
    """bseq_read_orig (src/bwa.cpp:266-335): records until size >= chunk_size and an even count when paired"""
    w = loader.fastq_parse(text)
    assert w["status"] == 0
    lens = np.diff(w["cum"])
    cuts, size, n = [], 0, 0
    for i, l in enumerate(lens):
        size += int(l); n += 1
        if size >= chunk_bases and (not paired or n % 2 == 0):
            cuts.append(n); size = 0; n = 0
    if n:
        cuts.append(n)
    return w, cuts


@pytest.mark.parametrize("gz", [True, False])
def test_reader_cuts_chunks_where_bseq_read_orig_does(tmp_path, gz):
    from test_gpu_fastq import _wrapped_fastq
    from test_oracle_fastq import make_fastq
    capi.lib()
    texts = [make_fastq(3000, 5)[0].replace(b"-", b"N"), _wrapped_fastq(2500, 6), _wrapped_fastq(700, 7, crlf=True)]
    for k, text in enumerate(texts):
        path = str(tmp_path / (f"r{k}.fq.gz" if gz else f"r{k}.fq"))
        (gzip.open if gz else open)(path, "wb").write(text)
        for chunk_bases, paired in ((40000, False), (100000, True), (10 ** 9, False), (1, True)):
            w, cuts = _expected_cuts(text, chunk_bases, paired)
            got = _reader_chunks(path, chunk_bases, paired)
            assert [g[1] for g in got] == cuts, (k, chunk_bases, paired)
            assert b"".join(g[0] for g in got).replace(b"\r", b"").replace(b"\n", b"") == text.replace(b"\r", b"").replace(b"\n", b"")
            at = 0
            for t, nr, nbases in got:                      # every chunk parses on its own to its slice of the records
                sub = loader.fastq_parse(t)
                assert sub["n"] == nr and sub["names"] == w["names"][at:at + nr] and int(sub["cum"][-1]) == nbases
                at += nr
    # a file that ends inside a record, and text that is no record
    bad = str(tmp_path / "bad.fq")
    open(bad, "wb").write(b"@a\nACGT\n+\nIIII\n@b\nACGT\n+\nII")          # (a last record without '+' would be a FASTA record to the reader)
    L = capi.lib()
    r = C.c_void_p()
    capi._chk(L.bwams_reader_open(bad.encode(), C.c_int64(100), 0, C.c_int64(0), 2, C.byref(r)), "open")
    text, nb, nr, nbs = C.c_void_p(), C.c_int64(0), C.c_int64(0), C.c_int64(0)
    rcs = []
    for _ in range(3):
        rcs.append(L.bwams_reader_next(r, C.byref(text), C.byref(nb), C.byref(nr), C.byref(nbs)))
        if rcs[-1] != 0:
            break
        L.bwams_reader_release(r, text)
    assert rcs[-1] < 0
    L.bwams_reader_close(r)


def test_bseq_parse_in_place_equals_the_reader_of_the_reference():
    from test_gpu_fastq import _wrapped_fastq
    from test_oracle_fastq import make_fastq
    parse = capi._host_sym("bwams_bseq_parse")
    parse.restype = C.c_int64
    for text in (make_fastq(2000, 9)[0].replace(b"-", b"N"), _wrapped_fastq(1500, 10), _wrapped_fastq(500, 11, crlf=True),
                 b">fa1 x y\nACGT\nAC\n>fa2\n\nGG\n"):
        w = loader.fastq_parse(text)
        buf = C.create_string_buffer(text, len(text) + 1)
        seqs = np.zeros(w["n"] + 1, capi.BSEQ1_DTYPE)
        for copy_comment in (1, 0):
            buf = C.create_string_buffer(text, len(text) + 1)
            n = parse(buf, C.c_int64(len(text)), C.c_int64(w["n"]), C.c_void_p(seqs.ctypes.data), copy_comment)
            assert n == w["n"]
            for i in range(n):
                s = seqs[i]
                assert C.string_at(int(s["name"])) == w["names"][i]
                assert int(s["l_seq"]) == int(w["cum"][i + 1] - w["cum"][i])
                sq = C.string_at(int(s["seq"]))
                enc = np.array([capi_nt4(c) for c in sq], np.uint8) if sq else np.zeros(0, np.uint8)
                assert np.array_equal(enc, w["enc"][w["cum"][i]:w["cum"][i + 1]]), i
                if w["has_qual"][i]:
                    assert C.string_at(int(s["qual"])) == w["quals"][w["cum"][i]:w["cum"][i + 1]].tobytes()
                else:
                    assert int(s["qual"]) == 0
                cm = w["comments"][i] if copy_comment else None
                assert (C.string_at(int(s["comment"])) if int(s["comment"]) else None) == cm


def capi_nt4(c):
    return {65: 0, 97: 0, 67: 1, 99: 1, 71: 2, 103: 2, 84: 3, 116: 3, 45: 5}.get(c, 4)


def test_writer_shards_are_written_in_sequence_order(tmp_path):
    L = capi.lib()
    w = C.c_void_p()
    capi._chk(L.bwams_writer_open(str(tmp_path / "out").encode(), 3, C.byref(w)), "bwams_writer_open")
    rng = np.random.default_rng(3)
    want = {s: [] for s in range(3)}
    items = [(s, q, (b"shard%d chunk%d\n" % (s, q)) * int(rng.integers(0, 2000))) for s in range(3) for q in range(40)]
    for s, q, t in items:
        want[s].append(t)
    for i in rng.permutation(len(items)):
        s, q, t = items[i]
        capi._chk(L.bwams_writer_put(w, s, C.c_int64(q), t, C.c_int64(len(t))), "bwams_writer_put")
    assert L.bwams_writer_put(w, 0, C.c_int64(5), b"x", C.c_int64(1)) != 0                # a sequence number twice
    capi._chk(L.bwams_writer_close(w), "bwams_writer_close")
    for s in range(3):
        assert open(tmp_path / f"out.{s}.sam", "rb").read() == b"".join(want[s])
    w = C.c_void_p()
    capi._chk(L.bwams_writer_open(str(tmp_path / "one.sam").encode(), 1, C.byref(w)), "bwams_writer_open")
    for q in (2, 0, 1):
        capi._chk(L.bwams_writer_put(w, 0, C.c_int64(q), b"%d\n" % q, C.c_int64(2)), "put")
    capi._chk(L.bwams_writer_close(w), "close")
    assert open(tmp_path / "one.sam", "rb").read() == b"0\n1\n2\n"


def test_gz_file_streamed_through_mem_process_seqs(tmp_path):
    """A .gz FASTQ file -> bwams_reader (inflate + cut on its own thread) -> bwams_bseq_parse (step 0's records, in place) ->
    mem_process_seqs_stage / mem_process_seqs / mem_process_seqs_collect over two chunks in flight -> bwams_writer: the file written
    equals bwams_process_chunk over the same chunks (the reads' ordinals carried), wrapped records and comments included."""
    from test_host_boundary import _setup
    g, ix, contigs, cnames = _setup(seed=29)
    reads, _, _ = simulate.make_reads(g, 5000, seed=71)
    rng = np.random.default_rng(5)
    recs = []
    for i, r in enumerate(reads):
        sq = bytes(b"ACGTN"[c] for c in r)
        q = bytes(rng.integers(35, 74, size=len(sq), dtype=np.uint8))
        if i % 9 == 0:                                       # a wrapped record
            recs.append(b"@g%d some comment\n%s\n%s\n+\n%s\n%s\n" % (i, sq[:70], sq[70:], q[:100], q[100:]))
        else:
            recs.append(b"@g%d/1\n%s\n+\n%s\n" % (i, sq, q))
    text = b"".join(recs)
    path = str(tmp_path / "reads.fq.gz")
    gzip.open(path, "wb").write(text)
    chunk_bases = 150 * 1200
    L = capi.lib()
    parse = capi._host_sym("bwams_bseq_parse")
    parse.restype = C.c_int64
    opt = capi.mem_opt_init(False)
    wk = capi.Worker([ix], 2000, 2000 * 160, depth=2)
    wk.set_deferred_collect(True)
    rd, wr = C.c_void_p(), C.c_void_p()
    capi._chk(L.bwams_reader_open(path.encode(), C.c_int64(chunk_bases), 0, C.c_int64(0), 3, C.byref(rd)), "bwams_reader_open")
    capi._chk(L.bwams_writer_open(str(tmp_path / "out.sam").encode(), 1, C.byref(wr)), "bwams_writer_open")
    chunks, n_done, pending = [], 0, []

    class S:                                                 # a chunk between the steps: the bseq1_t array over the reader's buffer
        pass

    def collect(s, k):
        wk.collect(opt, s)
        t = s.take_sam()
        capi._chk(L.bwams_writer_put(wr, 0, C.c_int64(k), t, C.c_int64(len(t))), "bwams_writer_put")
        capi._chk(L.bwams_reader_release(rd, s.text), "bwams_reader_release")

    k = 0
    while True:
        tp, nb, nr, nbs = C.c_void_p(), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        rc = L.bwams_reader_next(rd, C.byref(tp), C.byref(nb), C.byref(nr), C.byref(nbs))
        if rc == 1:
            break
        assert rc == 0
        chunks.append((C.string_at(tp.value, nb.value), nr.value))
        s = S()
        s.text, s.n = tp, nr.value
        s.arr = np.zeros(nr.value, capi.BSEQ1_DTYPE)
        s.ptr = C.c_void_p(s.arr.ctypes.data)
        s.take_sam = capi.Seqs.take_sam.__get__(s)
        assert parse(tp, nb, nr, s.ptr, 0) == nr.value
        wk.stage(opt, s)                                      # chunk k goes up ...
        if pending:
            ps, pk, pn = pending.pop()
            wk.process(opt, pn, ps)                           # ... while chunk k - 1 computes
            collect(ps, pk)
        pending.append((s, k, n_done))
        n_done += nr.value
        k += 1
    ps, pk, pn = pending.pop()
    wk.process(opt, pn, ps)
    collect(ps, pk)
    capi._chk(L.bwams_writer_close(wr), "bwams_writer_close")
    L.bwams_reader_close(rd)
    wk.close()
    assert sum(c[1] for c in chunks) == len(reads) and len(chunks) >= 4
    b = capi.Batch(ix, 2000, 2000 * 160)
    want, first = b"", 0
    for t, nr in chunks:
        sam, _ = b.process_chunk(t, n_processed=first)
        want += sam
        first += nr
    b.close()
    got = open(tmp_path / "out.sam", "rb").read()
    assert got == want and got.count(b"\n") >= len(reads)
    ix.close()
