"""The compiled caller at the reference's outer boundary (bwa-mem-scale_amd/host/): mem_process_seqs() with the signature of
src/bwamem.h:393-395 over layout mirrors of mem_opt_t / bseq1_t / mem_pestat_t, and the N-batches-behind-one-call driver
(host/chunk_multi.cpp).  CPU part: the driver compiles and links against libbwams.so, the option mapping equals the library's
defaults (= mem_opt_init), a missing index / GPU ends the run the reference's way, and the shard arithmetic.  GPU part: the
driver's SAM file == bwams_process_chunk's bytes (single-end and paired-end, several chunks, 512-read work items), and two
batches on one device reproduce the single-batch text byte for byte."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from bwams import capi, fmindex, shard, simulate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "bwa-mem-scale_amd", "_build")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    capi.build()
    exe = str(tmp_path_factory.mktemp("host") / "host_driver")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "bwa-mem-scale_amd", "host"), os.path.join(ROOT, "tests", "host_driver.cpp"), "-o", exe,
                           "-L" + BUILD, "-lbwams", "-Wl,-rpath," + BUILD])
    return exe


def test_option_mapping_equals_the_library_defaults(driver):
    out = subprocess.check_output([driver, "options"]).decode().split("\n")
    so, mo, sa = capi.default_seed_opt(), capi.default_mem_opt(), capi.default_sam_opt()
    assert out[0].split() == ["seed", str(so.min_seed_len), "%.3f" % so.split_factor, str(so.split_width), str(so.max_mem_intv), str(so.max_occ)]
    want = [mo.a, mo.b, mo.o_del, mo.e_del, mo.o_ins, mo.e_ins, mo.pen_clip5, mo.pen_clip3, mo.w, mo.zdrop, mo.min_seed_len, mo.min_chain_weight,
            mo.max_chain_extend, mo.max_occ, mo.max_chain_gap, "%.3f" % mo.mask_level, "%.3f" % mo.drop_ratio, "%.3f" % mo.mask_level_redun,
            mo.max_ins, mo.pen_unpaired, mo.max_matesw, mo.mapq_coef_len]
    assert out[1].split() == ["mem"] + [str(x) for x in want]
    assert [int(x) for x in out[2].split()[1:]] == list(mo.mat)
    assert out[3].split() == ["sam", str(sa.T), str(sa.flag), "%.3f" % sa.XA_drop_ratio, str(sa.max_XA_hits), str(sa.max_XA_hits_alt)]


def test_missing_index_ends_the_run_like_the_reference(driver, tmp_path):
    r = subprocess.run([driver, "run", str(tmp_path / "nothing"), "x.fq", "o.sam", "se", "100", "c:0:10:0"], capture_output=True)
    assert r.returncode == 1 and b"[bwams]" in r.stderr and b"cannot open" in r.stderr


def test_shard_bounds_arithmetic():
    rng = np.random.default_rng(1)
    for _ in range(300):
        n, w, pe = int(rng.integers(0, 5000)), int(rng.integers(1, 9)), bool(rng.integers(0, 2))
        n -= n & 1 if pe else 0
        b = capi.shard_bounds(n, w, pe)
        assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) >= 0)
        unit = 2 if pe else 1
        sizes = np.diff(b) // unit
        assert np.all(np.diff(b) % unit == 0) and sizes.max() - sizes.min() <= 1 and np.all(np.diff(sizes) <= 0)
        assert np.array_equal(b, shard.shard_bounds(n, w, unit))        # the Python ranks cut the same way
    lib = capi.lib()
    assert lib.bwams_shard_bounds(C.c_int64(7), 2, 1, capi._p(np.zeros(3, np.int64))) != 0      # an odd paired chunk is refused


# ------------------------------------------------------------------------------------------------------------------ GPU
def _fastq(reads, names, quals=None):
    out = []
    for i, r in enumerate(reads):
        s = "".join("ACGTN"[c] for c in r)
        q = quals[i] if quals is not None else "I" * len(s)
        out.append(f"@{names[i]}\n{s}\n+\n{q}\n")
    return "".join(out).encode()


def _setup(n_bases=300000, seed=21):
    g = simulate.make_genome(n_bases, seed=seed, repeat_frac=0.2, repeat_len=250, n_families=4)
    ix = capi.Index.build(g, 0)
    contigs = np.zeros(3, capi.CONTIG_DTYPE)
    contigs["offset"], contigs["len"], contigs["is_alt"] = [0, 90000, 200000], [90000, 110000, n_bases - 200000], [0, 0, 1]
    ix.set_contigs(contigs)
    names = ["chrA", "chrB", "chrC_alt"]
    ix.set_contig_names(names)
    return g, ix, contigs, names


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["se", "pe"])
def test_compiled_caller_reproduces_process_chunk(driver, tmp_path, mode):
    g, ix, contigs, cnames = _setup()
    prefix = str(tmp_path / "toy")
    ix.save(prefix)
    if mode == "se":
        reads, _, _ = simulate.make_reads(g, 2600, seed=5)
        reads = [np.array(r, np.uint8) for r in reads]
        reads[7][3] = 4
        names = [f"r{i}" for i in range(len(reads))]
    else:
        pr = simulate.make_read_pairs(g, 1300, seed=6, read_len=150, insert_mean=400.0, insert_sd=30.0, damaged_frac=0.2, discordant_frac=0.05)
        reads = [np.array(r, np.uint8) for r in pr]
        names = [f"p{i // 2}" for i in range(len(reads))]
    rng = np.random.default_rng(3)
    quals = ["".join(chr(int(x)) for x in rng.integers(35, 74, size=len(r))) for r in reads]
    fq = _fastq(reads, names, quals)
    fqp = tmp_path / "reads.fq"
    fqp.write_bytes(fq)
    chunk = 1100 if mode == "se" else 1000                      # several chunks, each of several 512-read work items
    spec = ",".join(f"{n}:{int(c['offset'])}:{int(c['len'])}:{int(c['is_alt'])}" for n, c in zip(cnames, contigs))
    out = tmp_path / "out.sam"
    r = subprocess.run([driver, "run", prefix, str(fqp), str(out), mode, str(chunk), spec], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    # the same chunks through the text-to-text entry point of the library
    want = b""
    b = capi.Batch(ix, chunk, chunk * 400)
    for first in range(0, len(reads), chunk):
        part = _fastq(reads[first:first + chunk], names[first:first + chunk], quals[first:first + chunk])
        text, off = b.process_chunk(part, paired=(mode == "pe"), n_processed=first)
        want += text
    b.close()
    got = out.read_bytes()
    assert got == want and got.count(b"\n") >= len(reads)
    ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["se", "pe"])
def test_two_batches_behind_one_call_equal_one_batch(mode):
    g, ix, contigs, cnames = _setup(seed=22)
    if mode == "se":
        reads, _, _ = simulate.make_reads(g, 3001, seed=8)
        reads = [np.array(r, np.uint8) for r in reads]
        names = [f"read{i}" for i in range(len(reads))]
    else:
        pr = simulate.make_read_pairs(g, 1501, seed=9, read_len=150, insert_mean=380.0, insert_sd=35.0, damaged_frac=0.25, discordant_frac=0.06)
        reads = [np.array(r, np.uint8) for r in pr]
        names = [f"pair{i // 2}" for i in range(len(reads))]
    enc, cum = simulate.flatten_reads(reads)
    nm = np.frombuffer("".join(names).encode(), np.uint8)
    noff = np.concatenate([[0], np.cumsum([len(x) for x in names])]).astype(np.int64)
    rng = np.random.default_rng(4)
    quals = rng.integers(35, 74, size=int(cum[-1])).astype(np.uint8)
    ID0 = 123456
    one = capi.Batch(ix, len(reads), int(cum[-1]))
    want, woff = one.process_reads(enc, cum, nm, noff, quals=quals, paired=(mode == "pe"), n_processed=ID0)
    one.close()
    for n_shards in (2, 3):
        bs = [capi.Batch(ix, len(reads), int(cum[-1])) for _ in range(n_shards)]
        m = capi.Multi(bs)
        got, goff = m.process_reads(enc, cum, nm, noff, quals=quals, paired=(mode == "pe"), n_processed=ID0)
        m.close()
        for b in bs:
            b.close()
        assert got == want and np.array_equal(goff, woff), n_shards
    # the parsed-records entry point equals the text entry point on the same chunk
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    text, off = b.process_chunk(_fastq(reads, names, ["".join(chr(int(c)) for c in quals[cum[i]:cum[i + 1]]) for i in range(len(reads))]),
                                paired=(mode == "pe"), n_processed=ID0)
    b.close()
    assert text == want and np.array_equal(off, woff)
    ix.close()


def _names_blob(fmt, ids):
    names = [fmt % int(i) for i in ids]
    return np.frombuffer(b"".join(names), np.uint8), np.concatenate([[0], np.cumsum([len(x) for x in names])]).astype(np.int64)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["se", "pe"])
def test_worker_pipeline_equals_chunk_by_chunk(mode):
    """The three steps of kt_pipeline over the compiled boundary (host/mem_process_seqs_hip.cpp driven through bwams/stream.py):
    mem_process_seqs_stage on the reader's thread, mem_process_seqs, mem_process_seqs_collect on the writer's — chunks in flight
    1, 2, 3; one and two batches per chunk (the chunk cut on read / pair boundaries); against mem_process_seqs alone (no overlap)
    and against bwams_process_reads chunk by chunk: the same bytes, the global read ordinals (hash seeds) carried through."""
    from bwams import stream
    g, ix, contigs, cnames = _setup(seed=23)
    paired = mode == "pe"
    n_chunks, per = 5, 1100 if not paired else 1000
    rng = np.random.default_rng(11)
    chunks = []
    for c in range(n_chunks):
        if not paired:
            rd, _, _ = simulate.make_reads(g, per, seed=50 + c)
            rd = np.asarray(rd, np.uint8)
            if c == 2:
                rd[5, 7] = 4
        else:
            rd = np.asarray(simulate.make_read_pairs(g, per // 2, seed=60 + c, read_len=150, insert_mean=400.0, insert_sd=30.0, damaged_frac=0.2,
                                                     discordant_frac=0.05), np.uint8)
        chunks.append(rd)
    first = np.concatenate([[0], np.cumsum([len(c) for c in chunks])])
    ID0 = 7000
    ids = [((ID0 + first[c] + np.arange(len(chunks[c]))) // 2 if paired else ID0 + first[c] + np.arange(len(chunks[c]))) for c in range(n_chunks)]
    fmt = b"q%08d"
    # the reference texts: one batch, chunk by chunk
    want = []
    b = capi.Batch(ix, per, per * 160)
    for c in range(n_chunks):
        enc, cum = simulate.flatten_reads(chunks[c])
        nm, noff = _names_blob(fmt, ids[c])
        text, _ = b.process_reads(enc, cum, nm, noff, quals=np.full(int(cum[-1]), ord("I"), np.uint8), paired=paired, n_processed=ID0 + int(first[c]))
        want.append(text)
    b.close()
    opt = capi.mem_opt_init(paired)

    def make(c):
        return capi.Seqs(chunks[c], name_fmt=fmt, name_ids=ids[c])

    for n_dev, depth, overlap in ((1, 1, False), (1, 1, True), (1, 2, True), (1, 3, True), (2, 2, True), (3, 3, True), (2, 1, False)):
        w = capi.Worker([ix] * n_dev, per, per * 160, depth=depth)
        got = {}
        secs, n = stream.run_job(w, opt, make, n_chunks, lambda i, t: got.__setitem__(i, t), n_processed0=ID0, overlap=overlap)
        w.close()
        assert n == first[-1]
        for c in range(n_chunks):
            assert got[c] == want[c], (n_dev, depth, overlap, c)
    # a chunk larger than the worker was sized for is an error code (and a message), not an exit
    w = capi.Worker([ix], 100, 100 * 160)
    with pytest.raises(capi.BwamsError, match="sized for"):
        w.process(opt, 0, make(0))
    w.close()
    ix.close()
