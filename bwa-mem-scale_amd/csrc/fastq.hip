// fastq.hip — read input on the device: a buffer of FASTQ text becomes the arrays the hot path and the SAM stage take.
//
// Replaces, for a buffer of whole records,
//   kseq_read                 /root/reference/src/kseq.h:358-400   (header '@', name up to the first isspace(), comment = rest of the
//                                                                   line, sequence, '+' line, quality; a line loses one trailing
//                                                                   '\r' when longer than one byte)
//   trim_readno, kseq2bseq1   /root/reference/src/bwa.cpp:74-153   ("/<digit>" name suffix; empty comment = none)
//   the base encoding         /root/reference/src/bwamem.cpp:1232  (seq[i] < 4 ? seq[i] : nst_nt4_table[seq[i]], bntseq.cpp:64-81)
// as bseq_read_orig (bwa.cpp:266-335) applies them per record.  kseq_read is a byte-serial state machine; its grammar also
// admits multi-line records, FASTA records and blank lines.  The device path takes the shape sequencers write — four lines per
// record — where every record is found from the line index alone, checks per record that the serial reader would have seen the
// same thing ('@' and '+' where they belong, equal sequence and quality lengths, no sequence line starting with '>', '+' or
// '@'), and returns BWAMS_ERR_UNSUPPORTED for any other input (the caller then reads that file on the host).  FASTA text (first byte
// '>', no line starting with '+' or '@') is taken too, sequences over any number of lines, blank lines skipped as the serial reader
// skips them: the header lines are found from the line index ('>' at a line start), a lane per record adds up its lines, a wave
// per record walks them again to encode.  Such a chunk has no qualities (bwams_fastq_has_qual = 0; the SAM text prints '*').
//
// Mapping.  HBM-bound streaming, three passes over the text: (1) the positions of the line ends (rocPRIM select over a counting
// iterator), (2) a lane per record: validate, measure name / comment / sequence, (3) after three exclusive scans, a WAVE per
// record copies name, comment and quality and encodes the bases, 64 bytes per step.  Algorithmic bytes: 3 reads of the text + 1
// write of ~0.9 of it.
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <memory>
#include <string>
#include <vector>
#include "common.h"

struct bwams_fastq {
    int device = 0;
    int64_t n_reads = 0, n_bases = 0, name_bytes = 0, comment_bytes = 0;
    void *d_enc = nullptr, *d_qual = nullptr, *d_names = nullptr, *d_comments = nullptr;
    std::vector<int64_t> cum, name_off, comment_off;       // host copies of the three offset arrays
    float ms = 0;
    bool has_qual = true;                                  // false: FASTA text
    // a decode that fails half way (an allocation, a kernel) drops the handle: whatever it had allocated goes with it
    ~bwams_fastq() {
        if (d_enc || d_qual || d_names || d_comments) (void)hipSetDevice(device);
        if (d_enc) (void)hipFree(d_enc);
        if (d_qual) (void)hipFree(d_qual);
        if (d_names) (void)hipFree(d_names);
        if (d_comments) (void)hipFree(d_comments);
    }
};

namespace bwams {
namespace {

struct IsLineEnd {
    const char *text;
    __device__ bool operator()(const int64_t &i) const { return text[i] == '\n'; }
};

__device__ __forceinline__ bool is_space(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }   // isspace(), "C" locale

__constant__ unsigned char kNt4[256] = {
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,5,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4, 4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4};

__global__ __launch_bounds__(256) void fastq_count_kernel(const char *__restrict__ text, int64_t n, unsigned long long *cnt) {
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) c += text[i] == '\n';
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(cnt, c);
}

struct Rec {                      // where the pieces of a record lie in the text
    int64_t name_at, comment_at, seq_at, qual_at;
    int32_t l_name, l_comment, l_seq;
    int32_t pad_;
};

// the header line [b, e): name up to the first isspace() (the delimiter decides whether a comment follows), trim_readno, comment
__device__ __forceinline__ void parse_header(const char *__restrict__ text, int64_t b, int64_t e, Rec &R) {
    int64_t p = b + 1;
    while (p < e && !is_space((unsigned char)text[p])) ++p;
    R.name_at = b + 1;
    int l_name = (int)(p - (b + 1));
    if (l_name > 2 && text[R.name_at + l_name - 2] == '/' && text[R.name_at + l_name - 1] >= '0' && text[R.name_at + l_name - 1] <= '9') l_name -= 2;
    R.l_name = l_name;
    R.comment_at = p < e ? p + 1 : e;
    int l_comment = p < e ? (int)(e - (p + 1)) : 0;
    if (l_comment > 1 && text[R.comment_at + l_comment - 1] == '\r') --l_comment;
    R.l_comment = l_comment;
}

// line j spans [start(j), end(j)): ends[j] is the position of its '\n' (or the text length for an unterminated last line)
__global__ void fastq_measure_kernel(const char *__restrict__ text, int64_t n_bytes, const int64_t *__restrict__ ends, int64_t n_lines,
                                     int64_t n_rec, Rec *__restrict__ rec, int64_t *__restrict__ wide, unsigned long long *bad) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rec) return;
    if (r == n_rec) { wide[r] = wide[n_rec + 1 + r] = wide[2 * (n_rec + 1) + r] = 0; return; }
    int64_t b[4], e[4];
    for (int k = 0; k < 4; ++k) {
        const int64_t j = 4 * r + k;
        b[k] = j ? ends[j - 1] + 1 : 0;
        e[k] = j < n_lines ? ends[j] : n_bytes;
    }
    bool ok = e[0] > b[0] && text[b[0]] == '@' && e[2] > b[2] && text[b[2]] == '+';
    Rec R;
    parse_header(text, b[0], e[0], R);
    int l_seq = (int)(e[1] - b[1]), l_qual = (int)(e[3] - b[3]);
    if (l_seq > 1 && text[e[1] - 1] == '\r') --l_seq;
    if (l_qual > 1 && text[e[3] - 1] == '\r') --l_qual;
    if (l_seq > 0) {
        const char c0 = text[b[1]];
        ok = ok && c0 != '>' && c0 != '+' && c0 != '@';
    }
    ok = ok && l_seq == l_qual;
    R.seq_at = b[1]; R.qual_at = b[3]; R.l_seq = l_seq; R.pad_ = 0;
    rec[r] = R;
    wide[r] = R.l_name; wide[n_rec + 1 + r] = R.l_comment; wide[2 * (n_rec + 1) + r] = l_seq;
    if (!ok) atomicAdd(bad, 1ull);
}

__global__ __launch_bounds__(256) void fastq_emit_kernel(const char *__restrict__ text, const Rec *__restrict__ rec, int64_t n_rec,
                                                         const int64_t *__restrict__ offs, char *__restrict__ names, char *__restrict__ comments,
                                                         uint8_t *__restrict__ enc, char *__restrict__ qual, unsigned long long *dash) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    unsigned long long n_dash = 0;
    for (int64_t r = wave; r < n_rec; r += n_waves) {
        const Rec R = rec[r];
        const int64_t no = offs[r], co = offs[n_rec + 1 + r], so = offs[2 * (n_rec + 1) + r];
        for (int i = lane; i < R.l_name; i += 64) names[no + i] = text[R.name_at + i];
        for (int i = lane; i < R.l_comment; i += 64) comments[co + i] = text[R.comment_at + i];
        for (int i = lane; i < R.l_seq; i += 64) {
            const unsigned char c = (unsigned char)text[R.seq_at + i];
            const unsigned char v = c < 4 ? c : kNt4[c];
            n_dash += v > 4;
            enc[so + i] = v;
            qual[so + i] = text[R.qual_at + i];
        }
    }
    if (__any(n_dash != 0) && n_dash) atomicAdd(dash, n_dash);
}

// ---- smart pairing (bseq_classify, bwa.cpp:346-362) and moving reads about ------------------------------------------------------
// eq[i] = read i carries the name of read i - 1 (strcmp == 0 on the trimmed names)
__global__ __launch_bounds__(256) void fastq_same_name_kernel(const char *__restrict__ names, const int64_t *__restrict__ name_off, int64_t n,
                                                             uint8_t *__restrict__ eq) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool same = false;
    if (i > 0) {
        const int64_t a = name_off[i - 1], b = name_off[i], l = b - a;
        same = name_off[i + 1] - b == l;
        for (int64_t k = 0; same && k < l; ++k) same = names[a + k] == names[b + k];
    }
    eq[i] = same;
}

// a wave per segment: len bytes from src to dst
__global__ __launch_bounds__(256) void segment_copy_kernel(const bwams::SegMove *__restrict__ mv, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t m = wave; m < n; m += n_waves) {
        const bwams::SegMove M = mv[m];
        for (int64_t i = lane; i < M.len; i += 64) M.dst[i] = M.src[i];
    }
}

// ---- FASTA text ------------------------------------------------------------------------------------------------------------
struct IsHeaderLine {                 // line j starts a record: its first byte is '>'
    const char *text;
    const int64_t *ends;
    int64_t n_nl, n_bytes;
    __device__ bool operator()(const int64_t &j) const {
        const int64_t b = j ? ends[j - 1] + 1 : 0, e = j < n_nl ? ends[j] : n_bytes;
        return e > b && text[b] == '>';
    }
};

// a line starting with '+' or '@' would send kseq_read into its FASTQ branch: not this path's input
__global__ __launch_bounds__(256) void fasta_check_kernel(const char *__restrict__ text, int64_t n_bytes, const int64_t *__restrict__ ends,
                                                          int64_t n_nl, int64_t n_lines, unsigned long long *bad) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_lines) return;
    const int64_t b = j ? ends[j - 1] + 1 : 0, e = j < n_nl ? ends[j] : n_bytes;
    if (e > b && (text[b] == '+' || text[b] == '@')) atomicAdd(bad, 1ull);
}

// one sequence line [b, e) appended to a sequence of l bytes (kseq_read's loop, kseq.h:377-381): the bytes it contributes.  An empty
// line is skipped; ks_getuntil_line2 drops a trailing '\r' when the WHOLE sequence so far is longer than one byte
__device__ __forceinline__ int fasta_line_bytes(const char *__restrict__ text, int64_t b, int64_t e, int l) {
    int len = (int)(e - b);
    if (len > 0 && l + len > 1 && text[e - 1] == '\r') --len;
    return len;
}

// lane per record: Rec.seq_at = index of its first sequence line, Rec.qual_at = index one past its last
__global__ void fasta_measure_kernel(const char *__restrict__ text, int64_t n_bytes, const int64_t *__restrict__ ends, int64_t n_nl, int64_t n_lines,
                                     const int64_t *__restrict__ hdr, int64_t n_rec, Rec *__restrict__ rec, int64_t *__restrict__ wide) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rec) return;
    if (r == n_rec) { wide[r] = wide[n_rec + 1 + r] = wide[2 * (n_rec + 1) + r] = 0; return; }
    const int64_t j0 = hdr[r], j1 = r + 1 < n_rec ? hdr[r + 1] : n_lines;
    Rec R;
    parse_header(text, j0 ? ends[j0 - 1] + 1 : 0, j0 < n_nl ? ends[j0] : n_bytes, R);
    int l = 0;
    for (int64_t j = j0 + 1; j < j1; ++j) {
        const int64_t b = ends[j - 1] + 1, e = j < n_nl ? ends[j] : n_bytes;
        l += fasta_line_bytes(text, b, e, l);
    }
    R.seq_at = j0 + 1; R.qual_at = j1; R.l_seq = l; R.pad_ = 0;
    rec[r] = R;
    wide[r] = R.l_name; wide[n_rec + 1 + r] = R.l_comment; wide[2 * (n_rec + 1) + r] = l;
}

// wave per record: names and comments as in fastq_emit_kernel; the sequence line by line, 64 bytes per step
__global__ __launch_bounds__(256) void fasta_emit_kernel(const char *__restrict__ text, int64_t n_bytes, const int64_t *__restrict__ ends, int64_t n_nl,
                                                         const Rec *__restrict__ rec, int64_t n_rec, const int64_t *__restrict__ offs,
                                                         char *__restrict__ names, char *__restrict__ comments, uint8_t *__restrict__ enc,
                                                         unsigned long long *dash) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    unsigned long long n_dash = 0;
    for (int64_t r = wave; r < n_rec; r += n_waves) {
        const Rec R = rec[r];
        const int64_t no = offs[r], co = offs[n_rec + 1 + r], so = offs[2 * (n_rec + 1) + r];
        for (int i = lane; i < R.l_name; i += 64) names[no + i] = text[R.name_at + i];
        for (int i = lane; i < R.l_comment; i += 64) comments[co + i] = text[R.comment_at + i];
        int l = 0;
        for (int64_t j = R.seq_at; j < R.qual_at; ++j) {
            const int64_t b = ends[j - 1] + 1, e = j < n_nl ? ends[j] : n_bytes;
            const int len = fasta_line_bytes(text, b, e, l);
            for (int i = lane; i < len; i += 64) {
                const unsigned char c = (unsigned char)text[b + i];
                const unsigned char v = c < 4 ? c : kNt4[c];
                n_dash += v > 4;
                enc[so + l + i] = v;
            }
            l += len;
        }
    }
    if (__any(n_dash != 0) && n_dash) atomicAdd(dash, n_dash);
}

// ---- FASTQ records over any number of lines ("wrapped" records) ---------------------------------------------------------------
// kseq_read's grammar (kseq.h:529-572) admits them: after the header, sequence lines until a line whose first byte is '+' (empty lines
// skipped), the rest of the '+' line ignored, then quality lines until the quality string is at least as long as the sequence.  A
// quality line may begin with '@' or '+', so records cannot be told from the line index alone: which lines are headers follows only from
// reading the records in order.  In parallel: EVERY line that begins with '@' is parsed as if a record began there (a lane each), which
// gives it a successor — the line where the next record would begin; the real records are the chain of successors from the first line,
// and record r of the chain is found by binary lifting over the successor table (r's bits select the jumps), a lane per record.
// Anything the serial reader would handle by scanning bytes for the next '@' or '>' (text between records, a '>' record in FASTQ
// text) ends the attempt with BWAMS_ERR_UNSUPPORTED; a truncated or overlong quality string is the reader's -2 and ends it too.
constexpr int kFqMaxLines = 1 << 20;          // a record of more lines than this is not parsed (nor is one the reader would accept)
constexpr int kFqLevels = 40;                 // successor tables for jumps of 2^0 .. 2^39 records
struct FqLine {
    int64_t b, e;                             // [b, e), the '\n' excluded
};
__device__ __forceinline__ FqLine fq_line(const int64_t *__restrict__ ends, int64_t n_nl, int64_t n_bytes, int64_t j) {
    FqLine L;
    L.b = j ? ends[j - 1] + 1 : 0;
    L.e = j < n_nl ? ends[j] : n_bytes;
    return L;
}
// the record whose header is line j: first sequence line, the '+' line, first quality line, one past the last quality line, lengths.
// Returns false when no record the reader accepts starts here.
struct FqShape {
    int64_t seq0, plus, qual0, qual1, next;   // line indices; next = the next record's header line, or n_lines at the end of the text
    int l_seq;
};
__device__ bool fq_shape(const char *__restrict__ text, int64_t n_bytes, const int64_t *__restrict__ ends, int64_t n_nl, int64_t n_lines,
                         int64_t j, FqShape &S) {
    int64_t jj = j + 1;
    int l = 0, steps = 0;
    S.seq0 = jj;
    for (;; ++jj) {
        if (jj >= n_lines || ++steps > kFqMaxLines) return false;          // no '+' line: a FASTA-like record, or the end of the text
        const FqLine L = fq_line(ends, n_nl, n_bytes, jj);
        if (L.e == L.b) continue;                                           // empty lines are skipped
        const char c0 = text[L.b];
        if (c0 == '+') break;
        if (c0 == '>' || c0 == '@') return false;                           // the next header before any '+': a record without qualities
        l += fasta_line_bytes(text, L.b, L.e, l);
    }
    S.plus = jj;
    S.l_seq = l;
    S.qual0 = ++jj;
    int lq = 0;
    do {                                                                    // at least one line, then until the quality is long enough
        if (jj >= n_lines || ++steps > kFqMaxLines) return false;           // truncated
        const FqLine L = fq_line(ends, n_nl, n_bytes, jj);
        lq += fasta_line_bytes(text, L.b, L.e, lq);                         // the same '\r' rule (ks_getuntil2 appends)
        ++jj;
    } while (lq < l);
    if (lq != l) return false;                                              // kseq_read's -2
    S.qual1 = jj;
    for (; jj < n_lines; ++jj) {                                            // the reader now scans for the next '@' or '>'
        const FqLine L = fq_line(ends, n_nl, n_bytes, jj);
        if (L.e == L.b || (L.e == L.b + 1 && text[L.b] == '\r')) continue;   // blank lines (of "\r\n" text too) hold neither
        if (text[L.b] != '@') return false;                                 // text between records, or a '>' record: not this path's input
        break;
    }
    S.next = jj;
    return true;
}
// up0[j] = successor of line j (n_lines: end of the text; n_lines + 1: no record starts at j)
__global__ __launch_bounds__(256) void fq_succ_kernel(const char *__restrict__ text, int64_t n_bytes, const int64_t *__restrict__ ends, int64_t n_nl,
                                                      int64_t n_lines, int64_t *__restrict__ up0) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n_lines + 1) return;
    int64_t nx = n_lines + 1;
    if (j >= n_lines) nx = j;                                               // the two terminals map to themselves
    else {
        const FqLine L = fq_line(ends, n_nl, n_bytes, j);
        FqShape S;
        if (L.e > L.b && text[L.b] == '@' && fq_shape(text, n_bytes, ends, n_nl, n_lines, j, S)) nx = S.next;
    }
    up0[j] = nx;
}
__global__ __launch_bounds__(256) void fq_double_kernel(const int64_t *__restrict__ a, int64_t *__restrict__ b, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) b[j] = a[a[j]];
}
// one thread: the first header line, and how many records the chain from it holds (-1: the chain runs into a line where no record starts)
__global__ void fq_count_kernel(const char *__restrict__ text, int64_t n_bytes, const int64_t *__restrict__ ends, int64_t n_nl, int64_t n_lines,
                                const int64_t *__restrict__ up, int levels, int64_t *__restrict__ out) {
    int64_t j = 0;
    for (; j < n_lines; ++j) {
        const FqLine L = fq_line(ends, n_nl, n_bytes, j);
        if (L.e > L.b && !(L.e == L.b + 1 && text[L.b] == '\r')) break;
    }
    out[0] = j;
    if (j >= n_lines) { out[1] = 0; return; }
    const int64_t stride = n_lines + 2;
    int64_t cur = j, n = 0;
    for (int k = levels - 1; k >= 0; --k) {
        const int64_t nx = up[(int64_t)k * stride + cur];
        if (nx < n_lines) { cur = nx; n += (int64_t)1 << k; }
    }
    // cur is the last header of the chain: its successor must be the end of the text
    out[1] = up[cur] == n_lines ? n + 1 : -1;
}
// lane per record: its header line by binary lifting, then the record's shape and lengths
__global__ void fq_wrapped_measure_kernel(const char *__restrict__ text, int64_t n_bytes, const int64_t *__restrict__ ends, int64_t n_nl, int64_t n_lines,
                                          const int64_t *__restrict__ up, int levels, int64_t first, int64_t n_rec, Rec *__restrict__ rec,
                                          int64_t *__restrict__ wide, unsigned long long *bad) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rec) return;
    if (r == n_rec) { wide[r] = wide[n_rec + 1 + r] = wide[2 * (n_rec + 1) + r] = 0; return; }
    const int64_t stride = n_lines + 2;
    int64_t cur = first;
    for (int k = 0; k < levels; ++k)
        if ((r >> k) & 1) cur = up[(int64_t)k * stride + cur];
    Rec R;
    R.name_at = R.comment_at = R.seq_at = R.qual_at = 0; R.l_name = R.l_comment = R.l_seq = 0; R.pad_ = 0;
    FqShape S;
    if (cur >= n_lines || !fq_shape(text, n_bytes, ends, n_nl, n_lines, cur, S)) {
        atomicAdd(bad, 1ull);
    } else {
        const FqLine H = fq_line(ends, n_nl, n_bytes, cur);
        parse_header(text, H.b, H.e, R);
        R.seq_at = S.seq0; R.qual_at = S.qual0; R.l_seq = S.l_seq;
        R.pad_ = (int32_t)(S.plus - S.seq0);                                // sequence lines (empty ones included)
    }
    rec[r] = R;
    wide[r] = R.l_name; wide[n_rec + 1 + r] = R.l_comment; wide[2 * (n_rec + 1) + r] = R.l_seq;
}
// wave per record: names and comments as in fastq_emit_kernel; sequence and quality line by line
__global__ __launch_bounds__(256) void fq_wrapped_emit_kernel(const char *__restrict__ text, int64_t n_bytes, const int64_t *__restrict__ ends, int64_t n_nl,
                                                              const Rec *__restrict__ rec, int64_t n_rec, const int64_t *__restrict__ offs,
                                                              char *__restrict__ names, char *__restrict__ comments, uint8_t *__restrict__ enc,
                                                              char *__restrict__ qual, unsigned long long *dash) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    unsigned long long n_dash = 0;
    for (int64_t r = wave; r < n_rec; r += n_waves) {
        const Rec R = rec[r];
        const int64_t no = offs[r], co = offs[n_rec + 1 + r], so = offs[2 * (n_rec + 1) + r];
        for (int i = lane; i < R.l_name; i += 64) names[no + i] = text[R.name_at + i];
        for (int i = lane; i < R.l_comment; i += 64) comments[co + i] = text[R.comment_at + i];
        int l = 0;
        for (int64_t j = R.seq_at; j < R.seq_at + R.pad_; ++j) {
            const FqLine L = fq_line(ends, n_nl, n_bytes, j);
            const int len = fasta_line_bytes(text, L.b, L.e, l);
            for (int i = lane; i < len; i += 64) {
                const unsigned char c = (unsigned char)text[L.b + i];
                const unsigned char v = c < 4 ? c : kNt4[c];
                n_dash += v > 4;
                enc[so + l + i] = v;
            }
            l += len;
        }
        int lq = 0;
        for (int64_t j = R.qual_at; lq < R.l_seq; ++j) {
            const FqLine L = fq_line(ends, n_nl, n_bytes, j);
            const int len = fasta_line_bytes(text, L.b, L.e, lq);
            for (int i = lane; i < len; i += 64) qual[so + lq + i] = text[L.b + i];
            lq += len;
        }
    }
    if (__any(n_dash != 0) && n_dash) atomicAdd(dash, n_dash);
}

}  // namespace
}  // namespace bwams

using namespace bwams;

namespace bwams {
int segment_copy(const std::vector<SegMove> &moves, hipStream_t st) {
    if (moves.empty()) return BWAMS_OK;
    SegMove *d = nullptr;
    BWAMS_HIP(dev_malloc(reinterpret_cast<void **>(&d), moves.size() * sizeof(SegMove)));
    hipError_t e = hipMemcpyAsync(d, moves.data(), moves.size() * sizeof(SegMove), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        int64_t blocks = ((int64_t)moves.size() + 3) / 4;
        if (blocks > 256 * 64) blocks = 256 * 64;
        segment_copy_kernel<<<(unsigned)blocks, 256, 0, st>>>(d, (int64_t)moves.size());
        e = hipStreamSynchronize(st);
    }
    (void)hipFree(d);
    BWAMS_HIP(e);
    return BWAMS_OK;
}

// bseq_classify (bwa.cpp:346-362) over the decoded chunk: which[i] = 1 when read i is an end of a pair (two neighbours of one name,
// taken greedily from the left), 0 when it stands alone
int fastq_classify(bwams_fastq *f, std::vector<uint8_t> *which) {
    const int64_t n = f->n_reads;
    which->assign((size_t)n, 0);
    if (n == 0) return BWAMS_OK;
    BWAMS_HIP(hipSetDevice(f->device));
    uint8_t *d_eq = nullptr;
    int64_t *d_off = nullptr;
    BWAMS_HIP(dev_malloc(reinterpret_cast<void **>(&d_eq), (size_t)n));
    hipError_t e = dev_malloc(reinterpret_cast<void **>(&d_off), (size_t)(n + 1) * 8);
    std::vector<uint8_t> eq((size_t)n);
    if (e == hipSuccess) e = hipMemcpy(d_off, f->name_off.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        fastq_same_name_kernel<<<(unsigned)((n + 255) / 256), 256, 0, nullptr>>>(reinterpret_cast<const char *>(f->d_names), d_off, n, d_eq);
        e = hipMemcpy(eq.data(), d_eq, (size_t)n, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_eq); (void)hipFree(d_off);
    BWAMS_HIP(e);
    int has_last = 1;                                    // the reference's loop, on the comparison results
    int64_t i;
    for (i = 1; i < n; ++i) {
        if (has_last) {
            if (eq[(size_t)i]) { (*which)[(size_t)i - 1] = (*which)[(size_t)i] = 1; has_last = 0; }
        } else has_last = 1;
    }
    return BWAMS_OK;
}

// the reads ids[0 .. n_ids) of f, in that order, as a chunk of their own
int fastq_subset(bwams_fastq *f, const std::vector<int64_t> &ids, bwams_fastq **out) {
    *out = nullptr;
    BWAMS_HIP(hipSetDevice(f->device));
    std::unique_ptr<bwams_fastq> g(new bwams_fastq());
    g->device = f->device; g->has_qual = f->has_qual;
    const size_t m = ids.size();
    g->cum.assign(m + 1, 0); g->name_off.assign(m + 1, 0); g->comment_off.assign(m + 1, 0);
    for (size_t k = 0; k < m; ++k) {
        const size_t r = (size_t)ids[k];
        g->cum[k + 1] = g->cum[k] + (f->cum[r + 1] - f->cum[r]);
        g->name_off[k + 1] = g->name_off[k] + (f->name_off[r + 1] - f->name_off[r]);
        g->comment_off[k + 1] = g->comment_off[k] + (f->comment_off[r + 1] - f->comment_off[r]);
    }
    g->n_reads = (int64_t)m; g->n_bases = g->cum[m]; g->name_bytes = g->name_off[m]; g->comment_bytes = g->comment_off[m];
    auto fail = [&](hipError_t e) { bwams_fastq *p = g.release(); bwams_fastq_close(p); return e; };
    hipError_t e = dev_malloc(&g->d_enc, (size_t)g->n_bases + 64);
    if (e == hipSuccess) e = dev_malloc(&g->d_qual, (size_t)g->n_bases + 64);
    if (e == hipSuccess) e = dev_malloc(&g->d_names, (size_t)g->name_bytes + 64);
    if (e == hipSuccess) e = dev_malloc(&g->d_comments, (size_t)g->comment_bytes + 64);
    if (e != hipSuccess) { BWAMS_HIP(fail(e)); }
    std::vector<SegMove> mv;
    mv.reserve(4 * m);
    for (size_t k = 0; k < m; ++k) {
        const size_t r = (size_t)ids[k];
        const char *se = reinterpret_cast<const char *>(f->d_enc) + f->cum[r], *sq = reinterpret_cast<const char *>(f->d_qual) + f->cum[r];
        mv.push_back({se, reinterpret_cast<char *>(g->d_enc) + g->cum[k], f->cum[r + 1] - f->cum[r]});
        if (f->has_qual) mv.push_back({sq, reinterpret_cast<char *>(g->d_qual) + g->cum[k], f->cum[r + 1] - f->cum[r]});
        mv.push_back({reinterpret_cast<const char *>(f->d_names) + f->name_off[r], reinterpret_cast<char *>(g->d_names) + g->name_off[k],
                      f->name_off[r + 1] - f->name_off[r]});
        if (f->comment_off[r + 1] > f->comment_off[r])
            mv.push_back({reinterpret_cast<const char *>(f->d_comments) + f->comment_off[r], reinterpret_cast<char *>(g->d_comments) + g->comment_off[k],
                          f->comment_off[r + 1] - f->comment_off[r]});
    }
    int rc = segment_copy(mv, nullptr);
    if (rc) { bwams_fastq *p = g.release(); bwams_fastq_close(p); return rc; }
    *out = g.release();
    return BWAMS_OK;
}

// bseq_read_orig with two files (bwa.cpp:275-318): read k of the first text, then read k of the second, for every k
int fastq_interleave(bwams_fastq *f1, bwams_fastq *f2, bwams_fastq **out) {
    *out = nullptr;
    if (f1->n_reads != f2->n_reads || f1->device != f2->device || f1->has_qual != f2->has_qual) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(f1->device));
    std::unique_ptr<bwams_fastq> g(new bwams_fastq());
    g->device = f1->device; g->has_qual = f1->has_qual;
    const size_t m = (size_t)f1->n_reads * 2;
    bwams_fastq *src[2] = {f1, f2};
    g->cum.assign(m + 1, 0); g->name_off.assign(m + 1, 0); g->comment_off.assign(m + 1, 0);
    for (size_t k = 0; k < m; ++k) {
        const bwams_fastq *f = src[k & 1];
        const size_t r = k >> 1;
        g->cum[k + 1] = g->cum[k] + (f->cum[r + 1] - f->cum[r]);
        g->name_off[k + 1] = g->name_off[k] + (f->name_off[r + 1] - f->name_off[r]);
        g->comment_off[k + 1] = g->comment_off[k] + (f->comment_off[r + 1] - f->comment_off[r]);
    }
    g->n_reads = (int64_t)m; g->n_bases = g->cum[m]; g->name_bytes = g->name_off[m]; g->comment_bytes = g->comment_off[m];
    hipError_t e = dev_malloc(&g->d_enc, (size_t)g->n_bases + 64);
    if (e == hipSuccess) e = dev_malloc(&g->d_qual, (size_t)g->n_bases + 64);
    if (e == hipSuccess) e = dev_malloc(&g->d_names, (size_t)g->name_bytes + 64);
    if (e == hipSuccess) e = dev_malloc(&g->d_comments, (size_t)g->comment_bytes + 64);
    if (e != hipSuccess) { bwams_fastq_close(g.release()); BWAMS_HIP(e); }
    std::vector<SegMove> mv;
    mv.reserve(4 * m);
    for (size_t k = 0; k < m; ++k) {
        const bwams_fastq *f = src[k & 1];
        const size_t r = k >> 1;
        const int64_t ls = f->cum[r + 1] - f->cum[r], ln = f->name_off[r + 1] - f->name_off[r], lc = f->comment_off[r + 1] - f->comment_off[r];
        mv.push_back({reinterpret_cast<const char *>(f->d_enc) + f->cum[r], reinterpret_cast<char *>(g->d_enc) + g->cum[k], ls});
        if (f->has_qual) mv.push_back({reinterpret_cast<const char *>(f->d_qual) + f->cum[r], reinterpret_cast<char *>(g->d_qual) + g->cum[k], ls});
        mv.push_back({reinterpret_cast<const char *>(f->d_names) + f->name_off[r], reinterpret_cast<char *>(g->d_names) + g->name_off[k], ln});
        if (lc) mv.push_back({reinterpret_cast<const char *>(f->d_comments) + f->comment_off[r], reinterpret_cast<char *>(g->d_comments) + g->comment_off[k], lc});
    }
    int rc = segment_copy(mv, nullptr);
    if (rc) { bwams_fastq_close(g.release()); return rc; }
    *out = g.release();
    return BWAMS_OK;
}
}  // namespace bwams

extern "C" {

int bwams_fastq_decode(int device, const char *text, int64_t n_bytes, bwams_fastq_t **out, int64_t *n_reads, int64_t *n_bases) {
    if (!text || n_bytes < 0 || !out) return BWAMS_ERR_ARG;
    *out = nullptr;
    BWAMS_HIP(hipSetDevice(device));
    hipStream_t st = nullptr;
    std::unique_ptr<bwams_fastq> f(new bwams_fastq());
    f->device = device;
    struct Events {                       // destroyed on every way out (the refusals return early)
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } evs;
    BWAMS_HIP(hipEventCreate(&evs.a)); BWAMS_HIP(hipEventCreate(&evs.b));
    const hipEvent_t e0 = evs.a, e1 = evs.b;
    // the text may already be in this GPU's memory (the copy kind is inferred)
    char *d_text = nullptr;
    hipPointerAttribute_t attr;
    const bool on_dev = hipPointerGetAttributes(&attr, text) == hipSuccess && attr.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    void *d_own = nullptr;
    if (on_dev) d_text = const_cast<char *>(text);
    else {
        BWAMS_HIP(dev_malloc(&d_own, (size_t)n_bytes + 16));
        d_text = reinterpret_cast<char *>(d_own);
        if (n_bytes) BWAMS_HIP(hipMemcpy(d_text, text, (size_t)n_bytes, hipMemcpyHostToDevice));
    }
    struct Scratch {
        std::vector<void *> p;
        ~Scratch() { for (void *q : p) if (q) (void)hipFree(q); }
    } scr;
    scr.p.push_back(d_own);
    BWAMS_HIP(hipEventRecord(e0, st));
    // (1) line ends
    int64_t *d_ends = nullptr, *d_cnt = nullptr;
    BWAMS_HIP(dev_malloc(reinterpret_cast<void **>(&d_cnt), 64)); scr.p.push_back(d_cnt);
    int64_t n_nl = 0;
    char last = '\n';
    if (n_bytes) {
        // count first: the array of line ends is sized exactly
        BWAMS_HIP(hipMemsetAsync(d_cnt, 0, 8, st));
        fastq_count_kernel<<<256 * 8, 256, 0, st>>>(d_text, n_bytes, reinterpret_cast<unsigned long long *>(d_cnt));
        BWAMS_HIP(hipMemcpyAsync(&n_nl, d_cnt, 8, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipStreamSynchronize(st));
        size_t tb = 0;
        rocprim::counting_iterator<int64_t> it(0);
        IsLineEnd pred{d_text};
        BWAMS_HIP(dev_malloc(reinterpret_cast<void **>(&d_ends), (size_t)(n_nl + 16) * 8)); scr.p.push_back(d_ends);
        BWAMS_HIP(rocprim::select(nullptr, tb, it, d_ends, d_cnt, (size_t)n_bytes, pred, st));
        void *d_tmp = nullptr;
        BWAMS_HIP(dev_malloc(&d_tmp, tb + 16)); scr.p.push_back(d_tmp);
        std::vector<char> tail(1);
        BWAMS_HIP(hipMemcpy(tail.data(), d_text + n_bytes - 1, 1, hipMemcpyDeviceToHost));
        last = tail[0];
        BWAMS_HIP(rocprim::select(d_tmp, tb, it, d_ends, d_cnt, (size_t)n_bytes, pred, st));
        BWAMS_HIP(hipMemcpyAsync(&n_nl, d_cnt, 8, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipStreamSynchronize(st));
    }
    const int64_t n_lines = n_nl + (n_bytes && last != '\n' ? 1 : 0);
    char first = '@';
    if (n_bytes) BWAMS_HIP(hipMemcpy(&first, d_text, 1, hipMemcpyDeviceToHost));
    const bool fasta = n_bytes && first == '>';
    unsigned long long *d_bad = nullptr;
    BWAMS_HIP(dev_malloc(reinterpret_cast<void **>(&d_bad), 64)); scr.p.push_back(d_bad);
    BWAMS_HIP(hipMemsetAsync(d_bad, 0, 16, st));
    int64_t n = 0;
    int64_t *d_hdr = nullptr;
    if (fasta) {
        // the records: the lines that start with '>'; no line may start with '+' or '@'
        fasta_check_kernel<<<(unsigned)((n_lines + 255) / 256), 256, 0, st>>>(d_text, n_bytes, d_ends, n_nl, n_lines, d_bad);
        BWAMS_HIP(dev_malloc(reinterpret_cast<void **>(&d_hdr), (size_t)(n_lines + 16) * 8)); scr.p.push_back(d_hdr);
        size_t tb = 0;
        rocprim::counting_iterator<int64_t> it(0);
        IsHeaderLine pred{d_text, d_ends, n_nl, n_bytes};
        BWAMS_HIP(rocprim::select(nullptr, tb, it, d_hdr, d_cnt, (size_t)n_lines, pred, st));
        void *d_tmp = nullptr;
        BWAMS_HIP(dev_malloc(&d_tmp, tb + 16)); scr.p.push_back(d_tmp);
        BWAMS_HIP(rocprim::select(d_tmp, tb, it, d_hdr, d_cnt, (size_t)n_lines, pred, st));
        unsigned long long bad0 = 0;
        BWAMS_HIP(hipMemcpyAsync(&n, d_cnt, 8, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipMemcpyAsync(&bad0, d_bad, 8, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipStreamSynchronize(st));
        if (bad0) {
            set_last_error("bwams_fastq_decode: FASTA text with " + std::to_string(bad0) + " line(s) starting with '+' or '@' (mixed or multi-line FASTQ input: read it on the host)");
            return BWAMS_ERR_UNSUPPORTED;
        }
    } else {
        n = n_lines / 4;
    }
    bool wrapped = !fasta && (n_lines % 4) != 0;        // not a whole number of four-line records: records over several lines, or nothing we take
    int64_t n1 = n + 1;
    // (2) measure + validate
    Rec *d_rec = nullptr;
    int64_t *d_wide = nullptr, *d_offs = nullptr;
    auto alloc_records = [&]() -> hipError_t {
        hipError_t e_ = dev_malloc(reinterpret_cast<void **>(&d_rec), (size_t)n1 * sizeof(Rec)); scr.p.push_back(d_rec);
        if (e_ == hipSuccess) { e_ = dev_malloc(reinterpret_cast<void **>(&d_wide), (size_t)n1 * 3 * 8); scr.p.push_back(d_wide); }
        if (e_ == hipSuccess) { e_ = dev_malloc(reinterpret_cast<void **>(&d_offs), (size_t)n1 * 3 * 8); scr.p.push_back(d_offs); }
        return e_;
    };
    if (!wrapped) {
        BWAMS_HIP(alloc_records());
        if (fasta) fasta_measure_kernel<<<(unsigned)((n1 + 255) / 256), 256, 0, st>>>(d_text, n_bytes, d_ends, n_nl, n_lines, d_hdr, n, d_rec, d_wide);
        else {
            fastq_measure_kernel<<<(unsigned)((n1 + 255) / 256), 256, 0, st>>>(d_text, n_bytes, d_ends, n_nl, n, d_rec, d_wide, d_bad);
            unsigned long long bad4 = 0;
            BWAMS_HIP(hipMemcpyAsync(&bad4, d_bad, 8, hipMemcpyDeviceToHost, st));
            BWAMS_HIP(hipStreamSynchronize(st));
            if (bad4) {                                 // some record is not on four lines: the general shape
                wrapped = true;
                BWAMS_HIP(hipMemsetAsync(d_bad, 0, 16, st));
            }
        }
    }
    if (wrapped) {
        // records over any number of lines (kseq_read's grammar): every '@' line parsed as if a record began there -> successor table ->
        // binary lifting from the first line
        int levels = 1;
        while (((int64_t)1 << levels) <= n_lines) ++levels;
        if (levels > kFqLevels) levels = kFqLevels;
        const int64_t stride = n_lines + 2;
        int64_t *d_up = nullptr, *d_fc = nullptr;
        BWAMS_HIP(dev_malloc(reinterpret_cast<void **>(&d_up), (size_t)levels * (size_t)stride * 8)); scr.p.push_back(d_up);
        BWAMS_HIP(dev_malloc(reinterpret_cast<void **>(&d_fc), 64)); scr.p.push_back(d_fc);
        const unsigned gb = (unsigned)((stride + 255) / 256);
        fq_succ_kernel<<<gb, 256, 0, st>>>(d_text, n_bytes, d_ends, n_nl, n_lines, d_up);
        for (int k = 1; k < levels; ++k) fq_double_kernel<<<gb, 256, 0, st>>>(d_up + (int64_t)(k - 1) * stride, d_up + (int64_t)k * stride, stride);
        fq_count_kernel<<<1, 1, 0, st>>>(d_text, n_bytes, d_ends, n_nl, n_lines, d_up, levels, d_fc);
        int64_t fc[2] = {0, 0};
        BWAMS_HIP(hipMemcpyAsync(fc, d_fc, 16, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipStreamSynchronize(st));
        if (fc[1] < 0) {
            set_last_error("bwams_fastq_decode: FASTQ text the device path does not take: text before the first '@' header or between records, a record "
                           "without a '+' line, or a quality string of another length than its sequence (read this input on the host)");
            return BWAMS_ERR_UNSUPPORTED;
        }
        n = fc[1];
        n1 = n + 1;
        if (d_rec) { d_rec = nullptr; d_wide = nullptr; d_offs = nullptr; }      // (the four-line attempt's arrays stay in scr until the end)
        BWAMS_HIP(alloc_records());
        fq_wrapped_measure_kernel<<<(unsigned)((n1 + 255) / 256), 256, 0, st>>>(d_text, n_bytes, d_ends, n_nl, n_lines, d_up, levels, fc[0], n, d_rec, d_wide, d_bad);
    }
    for (int row = 0; row < 3; ++row) {
        size_t tb = 0;
        BWAMS_HIP(rocprim::exclusive_scan(nullptr, tb, d_wide + row * n1, d_offs + row * n1, (int64_t)0, (size_t)n1, rocprim::plus<int64_t>(), st));
        void *d_tmp = nullptr;
        BWAMS_HIP(dev_malloc(&d_tmp, tb + 16)); scr.p.push_back(d_tmp);
        BWAMS_HIP(rocprim::exclusive_scan(d_tmp, tb, d_wide + row * n1, d_offs + row * n1, (int64_t)0, (size_t)n1, rocprim::plus<int64_t>(), st));
    }
    f->cum.resize((size_t)n1); f->name_off.resize((size_t)n1); f->comment_off.resize((size_t)n1);
    unsigned long long bad[2] = {0, 0};
    BWAMS_HIP(hipMemcpyAsync(f->name_off.data(), d_offs, (size_t)n1 * 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipMemcpyAsync(f->comment_off.data(), d_offs + n1, (size_t)n1 * 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipMemcpyAsync(f->cum.data(), d_offs + 2 * n1, (size_t)n1 * 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipMemcpyAsync(bad, d_bad, 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    if (bad[0]) {
        set_last_error("bwams_fastq_decode: " + std::to_string(bad[0]) + " record(s) are not '@' / sequence / '+' / quality of equal length (read this input on the host)");
        return BWAMS_ERR_UNSUPPORTED;
    }
    f->has_qual = !fasta;
    f->n_reads = n; f->n_bases = f->cum[(size_t)n]; f->name_bytes = f->name_off[(size_t)n]; f->comment_bytes = f->comment_off[(size_t)n];
    // (3) emit
    BWAMS_HIP(dev_malloc(&f->d_enc, (size_t)f->n_bases + 64));
    BWAMS_HIP(dev_malloc(&f->d_qual, (size_t)f->n_bases + 64));
    BWAMS_HIP(dev_malloc(&f->d_names, (size_t)f->name_bytes + 64));
    BWAMS_HIP(dev_malloc(&f->d_comments, (size_t)f->comment_bytes + 64));
    if (n) {
        int64_t blocks = (n + 3) / 4;
        if (blocks > 256 * 64) blocks = 256 * 64;
        if (fasta)
            fasta_emit_kernel<<<(unsigned)blocks, 256, 0, st>>>(d_text, n_bytes, d_ends, n_nl, d_rec, n, d_offs, reinterpret_cast<char *>(f->d_names),
                                                                reinterpret_cast<char *>(f->d_comments), reinterpret_cast<uint8_t *>(f->d_enc), d_bad + 1);
        else if (wrapped)
            fq_wrapped_emit_kernel<<<(unsigned)blocks, 256, 0, st>>>(d_text, n_bytes, d_ends, n_nl, d_rec, n, d_offs, reinterpret_cast<char *>(f->d_names),
                                                                     reinterpret_cast<char *>(f->d_comments), reinterpret_cast<uint8_t *>(f->d_enc),
                                                                     reinterpret_cast<char *>(f->d_qual), d_bad + 1);
        else
            fastq_emit_kernel<<<(unsigned)blocks, 256, 0, st>>>(d_text, d_rec, n, d_offs, reinterpret_cast<char *>(f->d_names),
                                                                reinterpret_cast<char *>(f->d_comments), reinterpret_cast<uint8_t *>(f->d_enc),
                                                                reinterpret_cast<char *>(f->d_qual), d_bad + 1);
    }
    BWAMS_HIP(hipEventRecord(e1, st));
    BWAMS_HIP(hipMemcpyAsync(bad + 1, d_bad + 1, 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    BWAMS_HIP(hipGetLastError());
    (void)hipEventElapsedTime(&f->ms, e0, e1);
    if (bad[1]) {
        set_last_error("bwams_fastq_decode: a '-' in a read (nst_nt4_table maps it to 5, outside the alphabet of the kernels)");
        return BWAMS_ERR_UNSUPPORTED;
    }
    if (n_reads) *n_reads = n;
    if (n_bases) *n_bases = f->n_bases;
    *out = f.release();
    return BWAMS_OK;
}

int bwams_fastq_info(const bwams_fastq_t *f, int64_t *n_reads, int64_t *n_bases, int64_t *name_bytes, int64_t *comment_bytes, float *ms) {
    if (!f) return BWAMS_ERR_ARG;
    if (n_reads) *n_reads = f->n_reads;
    if (n_bases) *n_bases = f->n_bases;
    if (name_bytes) *name_bytes = f->name_bytes;
    if (comment_bytes) *comment_bytes = f->comment_bytes;
    if (ms) *ms = f->ms;
    return BWAMS_OK;
}

int bwams_fastq_has_qual(const bwams_fastq_t *f) { return f && f->has_qual ? 1 : 0; }

int bwams_fastq_fetch(bwams_fastq_t *f, uint8_t *enc, int64_t *cum, char *names, int64_t *name_off, char *quals, char *comments,
                      int64_t *comment_off) {
    if (!f) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(f->device));
    const size_t n1 = (size_t)f->n_reads + 1;
    if (enc && f->n_bases) BWAMS_HIP(hipMemcpy(enc, f->d_enc, (size_t)f->n_bases, hipMemcpyDeviceToHost));
    if (quals && f->n_bases && f->has_qual) BWAMS_HIP(hipMemcpy(quals, f->d_qual, (size_t)f->n_bases, hipMemcpyDeviceToHost));
    if (names && f->name_bytes) BWAMS_HIP(hipMemcpy(names, f->d_names, (size_t)f->name_bytes, hipMemcpyDeviceToHost));
    if (comments && f->comment_bytes) BWAMS_HIP(hipMemcpy(comments, f->d_comments, (size_t)f->comment_bytes, hipMemcpyDeviceToHost));
    if (cum) memcpy(cum, f->cum.data(), n1 * 8);
    if (name_off) memcpy(name_off, f->name_off.data(), n1 * 8);
    if (comment_off) memcpy(comment_off, f->comment_off.data(), n1 * 8);
    return BWAMS_OK;
}

int bwams_fastq_to_batch_opt(bwams_fastq_t *f, bwams_batch_t *b, int32_t copy_comment) {
    if (!f || !b) return BWAMS_ERR_ARG;
    int rc = bwams_seed_upload(b, reinterpret_cast<const uint8_t *>(f->d_enc), f->cum.data(), nullptr, f->n_reads);
    if (rc) return rc;
    const bool cm = copy_comment && f->comment_bytes;
    return bwams_sam_upload(b, reinterpret_cast<const char *>(f->d_names), f->name_off.data(),
                            f->has_qual ? reinterpret_cast<const char *>(f->d_qual) : nullptr,
                            cm ? reinterpret_cast<const char *>(f->d_comments) : nullptr, cm ? f->comment_off.data() : nullptr);
}

int bwams_fastq_to_batch(bwams_fastq_t *f, bwams_batch_t *b) { return bwams_fastq_to_batch_opt(f, b, 1); }

int bwams_fastq_close(bwams_fastq_t *f) {
    delete f;                                              // the destructor frees the device arrays
    return BWAMS_OK;
}

}  // extern "C"
