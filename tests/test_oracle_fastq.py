"""The read-input step of the oracle (oracle/fastq_oracle.c: kseq_read + trim_readno + kseq2bseq1 + base encoding).  kseq.h is not
buildable here (safestringlib): PARITY UNPINNED; checked against an independent line parser and on the grammar's corners."""
import numpy as np

from oracle import loader

NT4 = {c: i for i, c in enumerate(b"ACGT")}
NT4.update({c: i for i, c in enumerate(b"acgt")})


def enc_of(s: bytes):
    return np.array([NT4.get(c, 5 if c == ord("-") else 4) for c in s], np.uint8)


def make_fastq(n, seed, crlf=False, comment_every=3, name_suffix=True, read_len=None):
    rng = np.random.default_rng(seed)
    recs, parts = [], []
    eol = b"\r\n" if crlf else b"\n"
    for i in range(n):
        L = int(read_len or rng.integers(1, 200))
        seq = bytes(rng.choice(list(b"ACGTNacgtnRY"), size=L, p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .01, .005, .005]).astype(np.uint8))
        qual = bytes(rng.integers(33, 75, size=L, dtype=np.uint8))
        name = b"read_%d" % i + ((b"/1" if i % 2 == 0 else b"/2") if name_suffix and i % 5 else b"")
        comment = (b"BC:Z:ACGT-%d extra words" % i) if comment_every and i % comment_every == 0 else b""
        sep = b" " if i % 2 else b"\t"
        parts.append(b"@" + name + (sep + comment if comment else b"") + eol + seq + eol + b"+" + (name if i % 7 == 0 else b"") + eol + qual + eol)
        nm = name[:-2] if len(name) > 2 and name[-2:-1] == b"/" and name[-1:].isdigit() else name
        recs.append((nm, comment or None, seq, qual))
    return b"".join(parts), recs


def check(got, recs):
    assert got["n"] == len(recs) and got["status"] == 0
    for i, (nm, cm, seq, qual) in enumerate(recs):
        assert got["names"][i] == nm and got["comments"][i] == cm, i
        a, b = got["cum"][i], got["cum"][i + 1]
        assert np.array_equal(got["enc"][a:b], enc_of(seq)), i
        assert bytes(got["quals"][a:b]) == qual and got["has_qual"][i] == (len(qual) > 0)


def test_four_line_fastq_equals_line_parser():
    for seed, crlf in ((1, False), (2, True), (3, False)):
        text, recs = make_fastq(400, seed, crlf=crlf)
        check(loader.fastq_parse(text), recs)
    text, recs = make_fastq(50, 4)
    check(loader.fastq_parse(text[:-1]), recs)                  # no newline at the end of the input


def test_grammar_corners():
    # multi-line sequence and quality, blank lines, text before the first header, a FASTA record in between
    text = (b"junk line\n@r1 c1\nACGT\nAC\n\n+\nIIII\nII\n>fa1 some comment\nACGTAC\nGT\n@r2/1\nNNAC-T\n+r2\n!!!!!!\n")
    g = loader.fastq_parse(text)
    assert g["n"] == 3 and g["names"] == [b"r1", b"fa1", b"r2"] and g["comments"] == [b"c1", b"some comment", None]
    assert list(g["cum"]) == [0, 6, 14, 20] and list(g["has_qual"]) == [1, 0, 1]
    assert list(g["enc"][14:20]) == [4, 4, 0, 1, 5, 3] and bytes(g["quals"][0:6]) == b"IIIIII"
    # a quality string of the wrong length stops the reader (kseq_read's -2)
    g = loader.fastq_parse(b"@a\nACGT\n+\nIIII\n@b\nAC\n+\nIIII\n@c\nAC\n+\nII\n")
    assert g["status"] == -2 and g["n"] == 1
    # a short quality line swallows the following lines until the length is reached, as kseq_read does
    g = loader.fastq_parse(b"@a\nACGT\n+\nII\n@c\nAC\n+\nII\n")
    assert g["status"] == 0 and g["n"] == 1 and bytes(g["quals"]) == b"II@c"
    # empty input, header only
    assert loader.fastq_parse(b"")["n"] == 0
    g = loader.fastq_parse(b"@only\n")
    assert g["n"] == 1 and g["names"] == [b"only"] and g["cum"][1] == 0
    # trim_readno needs more than two characters
    assert loader.fastq_parse(b"@/1\nA\n+\nI\n")["names"] == [b"/1"]
    assert loader.fastq_parse(b"@x/9\nA\n+\nI\n")["names"] == [b"x"]
