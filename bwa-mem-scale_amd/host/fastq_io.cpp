// fastq_io.cpp — the file ends of the pipeline around mem_process_seqs(): a reader thread that inflates a (gz or plain) FASTQ / FASTA
// file into page-locked chunk buffers cut where bseq_read_orig cuts its chunks, and a writer thread per output shard.
//
// Reference: kt_pipeline's step 0 is bseq_read_orig (/root/reference/src/bwa.cpp:266-335) over kseq_read on a gzFile — records are read
// until the chunk holds chunk_size BASES (and, for interleaved pairs, an even number of reads), one reader, inflate and parse on the
// same thread; step 2 is fputs() of every work item's string to ONE stream (src/fastmap.cpp:437-461).  Here:
//   * bwams_reader: a background thread inflates (zlib's gzread: gz and plain text alike) into a ring of page-locked buffers
//     (bwams_host_alloc: the chunk then goes up at the link's rate, or is parsed in place) and cuts chunks at the same records — the
//     records are found by lines with kseq_read's grammar (header; sequence lines up to the '+' line; quality lines until the quality is
//     as long as the sequence; '>' records without quality), so chunk i of this reader holds exactly the reads of chunk i of the
//     reference's with the same -K.  Reading chunk i + 1 overlaps whatever the caller does with chunk i.
//   * bwams_writer: one output stream per shard (per GPU: "<prefix>.<s>.sam", or a single file), each with a thread that writes the
//     texts handed to it in sequence-number order — a shard's file is in read order; concatenating the shards' files of a job that gave
//     shard s the s-th contiguous slice of every chunk is NOT read order across chunks (that is what the single-stream form is for).
// Host C++ only.  zlib is the reference's own dependency for this step (Makefile: -lz).
#include <zlib.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bwams.h"

namespace {

struct Chunk {
    char *buf = nullptr;          // page-locked, cap bytes
    int64_t cap = 0, n_bytes = 0, n_reads = 0, n_bases = 0;
    bool eof = false;
};

// One record of kseq_read's grammar starting at p (a '@' or '>' line) in [p, end): returns the position behind the record (the start
// of the next header, blank lines skipped) or -1 when the text ends inside the record (more input is needed); *bases = its l_seq.
// `final`: the input has ended, so a record that runs to the end of the text is complete.  -2: text the grammar does not start a
// record with (the serial reader would scan it byte by byte).
int64_t one_record(const char *t, int64_t p, int64_t end, bool final, int64_t *bases) {
    auto line_end = [&](int64_t a) -> int64_t {
        const void *q = memchr(t + a, '\n', (size_t)(end - a));
        return q ? (const char *)q - t : -1;
    };
    if (t[p] != '@' && t[p] != '>') return -2;
    int64_t e = line_end(p);
    if (e < 0) { if (!final) return -1; *bases = 0; return end; }
    int64_t a = e + 1, l_seq = 0;
    bool plus = false;
    while (a < end) {                                   // sequence lines
        const char c = t[a];
        if (c == '\n') { ++a; continue; }
        if (c == '+') { plus = true; break; }
        if (c == '>' || c == '@') break;
        e = line_end(a);
        int64_t le = e < 0 ? end : e;
        if (e < 0 && !final) return -1;
        int64_t len = le - a;
        if (len > 0 && l_seq + len > 1 && t[le - 1] == '\r') --len;
        l_seq += len;
        a = e < 0 ? end : e + 1;
    }
    if (a >= end && !final) return -1;                  // the next line may still belong to the record
    *bases = l_seq;
    if (!plus) return a;                                // FASTA-like record (or the end of the text)
    e = line_end(a);                                    // the '+' line
    if (e < 0) return final ? -2 : -1;
    a = e + 1;
    int64_t l_qual = 0;
    do {
        if (a >= end) return final ? -2 : -1;           // truncated quality (kseq_read's -2)
        e = line_end(a);
        if (e < 0 && !final) return -1;
        const int64_t le = e < 0 ? end : e;
        int64_t len = le - a;
        if (len > 0 && l_qual + len > 1 && t[le - 1] == '\r') --len;
        l_qual += len;
        a = e < 0 ? end : e + 1;
    } while (l_qual < l_seq);
    if (l_qual != l_seq) return -2;
    while (a < end && (t[a] == '\n' || (t[a] == '\r' && a + 1 < end && t[a + 1] == '\n'))) a += t[a] == '\n' ? 1 : 2;   // blank lines
    if (a >= end && !final) return -1;                  // (whether blank lines or the next header follow is not known yet)
    return a;
}

}  // namespace

struct bwams_reader {
    gzFile fp = nullptr;
    int64_t chunk_bases = 0;
    int paired = 0;
    std::vector<Chunk> ring;
    std::deque<int> free_q, ready_q;
    std::mutex mu;
    std::condition_variable cv;
    std::thread th;
    bool stop = false, done = false;
    int rc = BWAMS_OK;
    std::string err;
    // carry: bytes inflated but not yet part of a chunk
    std::vector<char> carry;
};

static void reader_main(bwams_reader *r) {
    bool file_end = false;
    for (;;) {
        int slot = -1;
        {
            std::unique_lock<std::mutex> g(r->mu);
            r->cv.wait(g, [r] { return r->stop || !r->free_q.empty(); });
            if (r->stop) break;
            slot = r->free_q.front();
            r->free_q.pop_front();
        }
        Chunk &c = r->ring[(size_t)slot];
        int64_t have = (int64_t)r->carry.size();
        if (have > c.cap) { r->rc = BWAMS_ERR_CAPACITY; r->err = "a record larger than the chunk buffer"; }
        if (have) memcpy(c.buf, r->carry.data(), (size_t)have);
        r->carry.clear();
        int64_t pos = 0, reads = 0, bases = 0;
        bool full = false;
        while (!full && !r->rc) {
            // parse what is there
            while (pos < have) {
                if (c.buf[pos] == '\n') { ++pos; continue; }                 // blank lines in front of a header
                int64_t b = 0;
                const int64_t nx = one_record(c.buf, pos, have, file_end, &b);
                if (nx == -1) break;
                if (nx == -2) { r->rc = BWAMS_ERR_UNSUPPORTED; r->err = "text that is not FASTQ / FASTA records near byte " + std::to_string(pos) + " of a chunk"; break; }
                pos = nx; ++reads; bases += b;
                if (bases >= r->chunk_bases && (!r->paired || (reads & 1) == 0)) { full = true; break; }     // bseq_read_orig's cut
            }
            if (full || r->rc || (file_end && pos >= have)) break;
            if (file_end) {                                                  // an incomplete last record
                if (pos < have) { r->rc = BWAMS_ERR_IO; r->err = "the file ends inside a record"; }
                break;
            }
            if (have >= c.cap - 1) { r->rc = BWAMS_ERR_CAPACITY; r->err = "chunk buffer too small for " + std::to_string(r->chunk_bases) + " bases of records"; break; }
            const int64_t want = std::min<int64_t>(c.cap - 1 - have, 8 << 20);      // one byte of slack: bwams_bseq_parse may put a NUL behind the text
            const int got = gzread(r->fp, c.buf + have, (unsigned)want);
            if (got < 0) { int e_ = 0; r->rc = BWAMS_ERR_IO; r->err = gzerror(r->fp, &e_); break; }
            if (got == 0) file_end = true;
            have += got;
        }
        c.n_bytes = pos; c.n_reads = reads; c.n_bases = bases;
        c.eof = file_end && pos >= have && reads == 0;
        if (!r->rc && have > pos) r->carry.assign(c.buf + pos, c.buf + have);
        {
            std::lock_guard<std::mutex> g(r->mu);
            r->ready_q.push_back(slot);
            if (r->rc || c.eof) r->done = true;
        }
        r->cv.notify_all();
        if (r->rc || c.eof) break;
    }
}

int bwams_reader_close(bwams_reader_t *r) {
    if (!r) return BWAMS_OK;
    {
        std::lock_guard<std::mutex> g(r->mu);
        r->stop = true;
    }
    r->cv.notify_all();
    if (r->th.joinable()) r->th.join();
    for (Chunk &c : r->ring) if (c.buf) bwams_host_free(c.buf);
    if (r->fp) gzclose(r->fp);
    delete r;
    return BWAMS_OK;
}

int bwams_reader_open(const char *path, int64_t chunk_bases, int32_t paired, int64_t buffer_bytes, int32_t n_buffers, bwams_reader_t **out) {
    if (!path || !out || chunk_bases <= 0 || n_buffers < 1 || n_buffers > 16) return BWAMS_ERR_ARG;
    *out = nullptr;
    bwams_reader *r = nullptr;
    try {
        r = new bwams_reader();
        r->fp = gzopen(path, "rb");
        if (!r->fp) { delete r; return BWAMS_ERR_IO; }
        gzbuffer(r->fp, 1 << 20);
        r->chunk_bases = chunk_bases;
        r->paired = paired;
        // a record of 150 bases is ~ 320 bytes of text: 2.6 bytes per base and room for the record that crosses the limit
        const int64_t cap = buffer_bytes > 0 ? buffer_bytes : chunk_bases * 3 + (64 << 20);
        r->ring.resize((size_t)n_buffers);
        for (int i = 0; i < n_buffers; ++i) {
            void *p = nullptr;
            if (int rc = bwams_host_alloc((size_t)cap, &p)) { bwams_reader_close(r); return rc; }
            r->ring[(size_t)i].buf = static_cast<char *>(p);
            r->ring[(size_t)i].cap = cap;
            r->free_q.push_back(i);
        }
        r->th = std::thread(reader_main, r);
    } catch (...) {
        if (r) bwams_reader_close(r);
        return BWAMS_ERR_NOMEM;
    }
    *out = r;
    return BWAMS_OK;
}

// The next chunk: its text (whole records, page-locked), bytes, reads and bases.  Returns 1 at the end of the file (no chunk), a
// negative code on a read / format error (bwams_reader_error).  The buffer stays the caller's until bwams_reader_release.
int bwams_reader_next(bwams_reader_t *r, const char **text, int64_t *n_bytes, int64_t *n_reads, int64_t *n_bases) {
    if (!r || !text) return BWAMS_ERR_ARG;
    int slot;
    {
        std::unique_lock<std::mutex> g(r->mu);
        r->cv.wait(g, [r] { return !r->ready_q.empty() || (r->done && r->ready_q.empty()); });
        if (r->ready_q.empty()) return r->rc ? r->rc : 1;
        slot = r->ready_q.front();
        r->ready_q.pop_front();
    }
    Chunk &c = r->ring[(size_t)slot];
    if (r->rc) return r->rc;
    if (c.eof) return 1;
    *text = c.buf;
    if (n_bytes) *n_bytes = c.n_bytes;
    if (n_reads) *n_reads = c.n_reads;
    if (n_bases) *n_bases = c.n_bases;
    return BWAMS_OK;
}

int bwams_reader_release(bwams_reader_t *r, const char *text) {
    if (!r || !text) return BWAMS_ERR_ARG;
    for (size_t i = 0; i < r->ring.size(); ++i)
        if (r->ring[i].buf == text) {
            {
                std::lock_guard<std::mutex> g(r->mu);
                r->free_q.push_back((int)i);
            }
            r->cv.notify_all();
            return BWAMS_OK;
        }
    return BWAMS_ERR_ARG;
}

const char *bwams_reader_error(const bwams_reader_t *r) { return r ? r->err.c_str() : ""; }

// ------------------------------------------------------------------------------------------------------------------------- writer
struct bwams_writer {
    struct Shard {
        FILE *fp = nullptr;
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::map<int64_t, std::string> pending;       // sequence number -> text (written when its turn comes)
        int64_t next = 0;
        bool stop = false;
        int rc = BWAMS_OK;
    };
    std::vector<Shard *> sh;
};

static void writer_main(bwams_writer::Shard *s) {
    for (;;) {
        std::string text;
        {
            std::unique_lock<std::mutex> g(s->mu);
            s->cv.wait(g, [s] { return s->stop || s->pending.count(s->next); });
            auto it = s->pending.find(s->next);
            if (it == s->pending.end()) { if (s->stop) return; continue; }
            text.swap(it->second);
            s->pending.erase(it);
            ++s->next;
        }
        if (!text.empty() && fwrite(text.data(), 1, text.size(), s->fp) != text.size()) s->rc = BWAMS_ERR_IO;
        s->cv.notify_all();
    }
}

int bwams_writer_close(bwams_writer_t *w) {
    if (!w) return BWAMS_OK;
    int rc = BWAMS_OK;
    for (auto *s : w->sh) {
        if (!s) continue;
        {
            std::unique_lock<std::mutex> g(s->mu);
            s->cv.wait(g, [s] { return s->pending.empty() || s->rc; });       // everything handed over in order has been written
            s->stop = true;
        }
        s->cv.notify_all();
        if (s->th.joinable()) s->th.join();
        if (s->fp && fclose(s->fp)) rc = BWAMS_ERR_IO;
        if (s->rc) rc = s->rc;
        delete s;
    }
    delete w;
    return rc;
}

// n_shards == 1: `path` is the file.  n_shards > 1: "<path>.<s>.sam", s = 0 .. n_shards - 1 (one per GPU).
int bwams_writer_open(const char *path, int32_t n_shards, bwams_writer_t **out) {
    if (!path || !out || n_shards < 1 || n_shards > 64) return BWAMS_ERR_ARG;
    *out = nullptr;
    bwams_writer *w = nullptr;
    try {
        w = new bwams_writer();
        for (int s = 0; s < n_shards; ++s) {
            auto *x = new bwams_writer::Shard();
            w->sh.push_back(x);
            const std::string name = n_shards == 1 ? std::string(path) : std::string(path) + "." + std::to_string(s) + ".sam";
            x->fp = fopen(name.c_str(), "wb");
            if (!x->fp) { bwams_writer_close(w); return BWAMS_ERR_IO; }
            setvbuf(x->fp, nullptr, _IOFBF, 8 << 20);
            x->th = std::thread(writer_main, x);
        }
    } catch (...) {
        if (w) bwams_writer_close(w);
        return BWAMS_ERR_NOMEM;
    }
    *out = w;
    return BWAMS_OK;
}

// Hand shard `shard` the text with sequence number `seq` (0, 1, 2 ... per shard, any arrival order); the bytes are copied, the call
// returns at once, the shard's thread writes seq 0, 1, 2 ... in that order.
int bwams_writer_put(bwams_writer_t *w, int32_t shard, int64_t seq, const char *text, int64_t n_bytes) {
    if (!w || shard < 0 || shard >= (int32_t)w->sh.size() || seq < 0 || n_bytes < 0 || (n_bytes && !text)) return BWAMS_ERR_ARG;
    auto *s = w->sh[(size_t)shard];
    try {
        std::string t(text ? text : "", (size_t)n_bytes);
        std::lock_guard<std::mutex> g(s->mu);
        if (seq < s->next || s->pending.count(seq)) return BWAMS_ERR_ARG;
        s->pending.emplace(seq, std::move(t));
    } catch (...) {
        return BWAMS_ERR_NOMEM;
    }
    s->cv.notify_all();
    return s->rc;
}

// ------------------------------------------------------------------------------------------------- step 0 for mem_process_seqs()
// bseq_read_orig's records from a chunk's text, IN PLACE: the strings of seqs[i] point into `text`, which is cut up with NULs
// (kseq2bseq1 strdup()s them, src/bwa.cpp:74-153; here the chunk buffer plays strbuf's part and lives until the chunk is written).
// Sequence and quality lines of wrapped records are joined in place.  copy_comment = 0: comments are dropped as process() drops them
// without `mem -C` (src/fastmap.cpp:356-363).  Returns the number of records (n_reads expected), or a negative code.
#include "bwamem_hip.h"

static inline bool is_space_c(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

int64_t bwams_bseq_parse(char *t, int64_t n_bytes, int64_t n_reads, bseq1_t *seqs, int copy_comment) {
    if (!t || !seqs || n_bytes < 0 || n_reads < 0) return BWAMS_ERR_ARG;
    const int64_t end = n_bytes;
    auto line_end = [&](int64_t a) -> int64_t {
        const void *q = a < end ? memchr(t + a, '\n', (size_t)(end - a)) : nullptr;
        return q ? (const char *)q - t : end;
    };
    int64_t p = 0, n = 0;
    while (p < end && n < n_reads) {
        if (t[p] == '\n' || t[p] == '\r') { ++p; continue; }
        if (t[p] != '@' && t[p] != '>') return BWAMS_ERR_UNSUPPORTED;
        bseq1_t &s = seqs[n];
        memset(&s, 0, sizeof s);
        int64_t e = line_end(p);
        // header: name up to the first isspace(), the rest of the line is the comment (one trailing '\r' dropped when longer than 1)
        int64_t q = p + 1;
        while (q < e && !is_space_c((unsigned char)t[q])) ++q;
        int64_t l_name = q - (p + 1);
        char *name = t + p + 1;
        int64_t l_comment = 0;
        char *comment = nullptr;
        if (q < e) { comment = t + q + 1; l_comment = e - (q + 1); if (l_comment > 1 && comment[l_comment - 1] == '\r') --l_comment; }
        if (l_name > 2 && name[l_name - 2] == '/' && name[l_name - 1] >= '0' && name[l_name - 1] <= '9') l_name -= 2;      // trim_readno
        int64_t a = e < end ? e + 1 : end;
        name[l_name] = 0;                                  // (the byte behind the name is the delimiter, a '/' or the line's '\n')
        if (comment) comment[l_comment] = 0;
        // sequence lines, joined at `dst`
        char *seq = t + a;
        int64_t l_seq = 0;
        bool plus = false;
        while (a < end) {
            const char c = t[a];
            if (c == '\n') { ++a; continue; }
            if (c == '+') { plus = true; break; }
            if (c == '>' || c == '@') break;
            e = line_end(a);
            int64_t len = e - a;
            if (len > 0 && l_seq + len > 1 && t[e - 1] == '\r') --len;
            if (seq + l_seq != t + a) memmove(seq + l_seq, t + a, (size_t)len);
            l_seq += len;
            a = e < end ? e + 1 : end;
        }
        char *qual = nullptr;
        if (plus) {
            e = line_end(a);
            a = e < end ? e + 1 : end;
            qual = t + a;
            int64_t l_qual = 0;
            do {
                if (a >= end) return BWAMS_ERR_IO;         // truncated quality: the reader's -2
                e = line_end(a);
                int64_t len = e - a;
                if (len > 0 && l_qual + len > 1 && t[e - 1] == '\r') --len;
                if (qual + l_qual != t + a) memmove(qual + l_qual, t + a, (size_t)len);
                l_qual += len;
                a = e < end ? e + 1 : end;
            } while (l_qual < l_seq);
            if (l_qual != l_seq) return BWAMS_ERR_IO;
            if (l_qual > 0) qual[l_qual] = 0;                    // lands on the last line's '\n' (or '\r'), never behind the record
            if (l_qual == 0) qual = nullptr;               // kseq2bseq1: an empty quality string is no quality string
        }
        // the NUL behind the sequence: the joined sequence ends at or before the '\n' of its last line (or at the '+' / next header when
        // the record has no bases: then that byte must stay, and an empty string is taken from the header line's own terminator)
        if (l_seq > 0) seq[l_seq] = 0; else seq = name + l_name;
        s.name = name;
        s.comment = (copy_comment && comment && l_comment > 0) ? comment : nullptr;
        s.seq = seq;
        s.qual = qual;
        s.l_seq = (int)l_seq;
        s.id = (int)n;
        ++n;
        p = a;
    }
    return n;
}
