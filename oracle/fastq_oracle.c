/*
 * fastq_oracle.c — CPU restatement of the read input step (TEST INFRASTRUCTURE ONLY, see bwams_oracle.h).
 *
 *   kseq_read                     /root/reference/src/kseq.h:358-400 (OPT_RW build; :530-572 is the same grammar)
 *   ks_getuntil_space / _line2    /root/reference/src/kseq.h:179-300  (name up to the first isspace(); a line loses one trailing '\r'
 *                                                                      when more than one byte long)
 *   trim_readno, kseq2bseq1       /root/reference/src/bwa.cpp:74-153   (a "/<digit>" suffix of the name goes; an empty comment or
 *                                                                      quality string becomes NULL)
 *   base encoding                 /root/reference/src/bwamem.cpp:1232, nst_nt4_table /root/reference/src/bntseq.cpp:64-81
 * over a memory buffer instead of a gzFile.
 *
 * PARITY UNPINNED: kseq.h includes memcpy_bwamem.h -> safestringlib (not buildable here).  Checked in tests/test_oracle_fastq.py
 * against an independent line-based parser on well-formed 4-line FASTQ, and on the grammar's corners (multi-line records, FASTA
 * records, blank lines, '\r\n', text before the first header, a truncated last record).
 */
#include <ctype.h>
#include <string.h>
#include "bwams_oracle.h"

typedef struct { const char *b; int64_t n, at; } ms_t;
static int ms_getc(ms_t *s) { return s->at < s->n ? (unsigned char)s->b[s->at++] : -1; }

/* ks_getuntil_line2 (append): bytes up to '\n' (consumed) are appended to out[*l ..]; returns -1 when nothing could be read at
 * end of input.  The '\r' rule looks at the whole string, as the original does. */
static int64_t get_line(ms_t *s, char *out, int64_t *l, int append)
{
    int gotany = 0;
    if (!append) *l = 0;
    if (s->at < s->n) {
        int64_t i = s->at;
        while (i < s->n && s->b[i] != '\n') ++i;
        memcpy(out + *l, s->b + s->at, (size_t)(i - s->at));
        *l += i - s->at;
        s->at = i < s->n ? i + 1 : i;
        gotany = 1;
    }
    if (!gotany) return -1;
    if (*l > 1 && out[*l - 1] == '\r') --*l;
    return *l;
}

static const unsigned char nt4[256] = {
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,5,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4, 4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4};

/* All records of buf.  names / comments / seq (encoded) / qual are written back to back; *_off hold n + 1 offsets (seq and qual
 * share cum: a record without qualities — FASTA — has has_qual = 0 and its qual bytes are left 0).  Every output buffer must hold
 * n bytes (offset arrays max_reads + 1 entries).  Returns the number of records read; -2 - (records read) when the reader stopped
 * at a quality string of the wrong length (kseq_read's -2). */
int64_t orc_fastq_parse(const char *buf, int64_t n, int64_t max_reads, char *names, int64_t *name_off, char *comments,
                        int64_t *comment_off, uint8_t *seq, char *qual, int64_t *cum, uint8_t *has_qual)
{
    ms_t s = {buf, n, 0};
    int last_char = 0;
    int64_t nr = 0, nl = 0, cl = 0, sl = 0;
    name_off[0] = comment_off[0] = cum[0] = 0;
    while (nr < max_reads) {
        int c;
        int64_t l_name = 0, l_comment = 0, l_seq = 0, l_qual = 0;
        char *nm = names + nl, *cm = comments + cl, *ql = qual + sl;
        char *sq = (char *)seq + sl;
        if (last_char == 0) {
            while ((c = ms_getc(&s)) != -1 && c != '>' && c != '@');
            if (c == -1) break;
            last_char = c;
        }
        {   /* ks_getuntil_space */
            int gotany = 0, d = 0;
            if (s.at < s.n) {
                int64_t i = s.at;
                while (i < s.n && !isspace((unsigned char)s.b[i])) ++i;
                memcpy(nm, s.b + s.at, (size_t)(i - s.at));
                l_name = i - s.at;
                if (i < s.n) d = (unsigned char)s.b[i];
                s.at = i < s.n ? i + 1 : i;
                gotany = 1;
            }
            if (!gotany) break;
            c = d;
        }
        if (c != '\n') { if (get_line(&s, cm, &l_comment, 0) < 0) l_comment = 0; }
        while ((c = ms_getc(&s)) != -1 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            sq[l_seq++] = (char)c;
            get_line(&s, sq, &l_seq, 1);
        }
        if (c == '>' || c == '@') last_char = c;
        has_qual[nr] = 0;
        if (c == '+') {
            while ((c = ms_getc(&s)) != -1 && c != '\n');
            if (c == -1) return -2 - nr;
            while (get_line(&s, ql, &l_qual, 1) >= 0 && l_qual < l_seq);
            last_char = 0;
            if (l_seq != l_qual) return -2 - nr;
            has_qual[nr] = l_qual > 0;
        } else if (c == -1) last_char = 0;
        /* trim_readno */
        if (l_name > 2 && nm[l_name - 2] == '/' && isdigit((unsigned char)nm[l_name - 1])) l_name -= 2;
        for (int64_t i = 0; i < l_seq; ++i) {
            const unsigned char b = (unsigned char)sq[i];
            seq[sl + i] = b < 4 ? b : nt4[b];
        }
        if (!has_qual[nr]) memset(ql, 0, (size_t)l_seq);
        nl += l_name; cl += l_comment; sl += l_seq;
        ++nr;
        name_off[nr] = nl; comment_off[nr] = cl; cum[nr] = sl;
        if (c == -1 && s.at >= s.n && last_char == 0) { /* end of input */ }
    }
    return nr;
}
