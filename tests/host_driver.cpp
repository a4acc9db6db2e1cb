// host_driver.cpp — a compiled caller at the reference's outer boundary, standing where kt_pipeline's step 1 stands
// (/root/reference/src/fastmap.cpp:392-419): records parsed on the host into bseq1_t, mem_process_seqs() with the reference's
// signature (bwa-mem-scale_amd/host/bwamem_hip.h), the work items' SAM strings written in order as step 2 does (:437-461).
// Test infrastructure: built and run by tests/test_host_boundary.py.
//   host_driver options                                    print the option records mem_opt_init()'s defaults map to
//   host_driver run <prefix> <fastq> <out.sam> <se|pe> <chunk_reads> name:offset:len:is_alt[,...]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bwamem_hip.h"

static void opt_init(mem_opt_t *o) {          // mem_opt_init, src/bwamem.cpp:135-171; bwa_fill_scmat, src/bwa.cpp
    memset(o, 0, sizeof *o);
    o->a = 1; o->b = 4; o->o_del = o->o_ins = 6; o->e_del = o->e_ins = 1; o->w = 100; o->T = 30; o->zdrop = 100;
    o->pen_unpaired = 17; o->pen_clip5 = o->pen_clip3 = 5; o->max_mem_intv = 20; o->min_seed_len = 19; o->split_width = 10;
    o->max_occ = 500; o->max_chain_gap = 10000; o->max_ins = 10000; o->mask_level = 0.50f; o->drop_ratio = 0.50f;
    o->XA_drop_ratio = 0.80f; o->split_factor = 1.5f; o->chunk_size = 10000000; o->n_threads = 1; o->max_XA_hits = 5;
    o->max_XA_hits_alt = 200; o->max_matesw = 50; o->mask_level_redun = 0.95f; o->min_chain_weight = 0; o->max_chain_extend = 1 << 30;
    o->mapQ_coef_len = 50; o->mapQ_coef_fac = (int)log(o->mapQ_coef_len);
    int k = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) o->mat[k++] = i == j ? o->a : -o->b;
        o->mat[k++] = -1;
    }
    for (int j = 0; j < 5; ++j) o->mat[k++] = -1;
}

int main(int argc, char **argv) {
    mem_opt_t opt;
    opt_init(&opt);
    if (argc >= 2 && !strcmp(argv[1], "options")) {
        bwams_seed_opt_t so; bwams_mem_opt_t mo; bwams_sam_opt_t sa;
        bwams_map_options(&opt, "", &so, &mo, &sa);
        printf("seed %d %.3f %d %d %d\n", so.min_seed_len, so.split_factor, so.split_width, so.max_mem_intv, so.max_occ);
        printf("mem %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %.3f %.3f %.3f %d %d %d %d\n", mo.a, mo.b, mo.o_del, mo.e_del, mo.o_ins, mo.e_ins,
               mo.pen_clip5, mo.pen_clip3, mo.w, mo.zdrop, mo.min_seed_len, mo.min_chain_weight, mo.max_chain_extend, mo.max_occ,
               mo.max_chain_gap, mo.mask_level, mo.drop_ratio, mo.mask_level_redun, mo.max_ins, mo.pen_unpaired, mo.max_matesw, mo.mapq_coef_len);
        printf("mat");
        for (int i = 0; i < 25; ++i) printf(" %d", mo.mat[i]);
        printf("\nsam %d %d %.3f %d %d\n", sa.T, sa.flag, sa.XA_drop_ratio, sa.max_XA_hits, sa.max_XA_hits_alt);
        return 0;
    }
    if (argc < 8 || strcmp(argv[1], "run")) { fprintf(stderr, "usage: host_driver run <prefix> <fastq> <out.sam> <se|pe> <chunk_reads> <contigs>\n"); return 2; }
    const char *prefix = argv[2], *fq = argv[3], *out = argv[4];
    const bool pe = !strcmp(argv[5], "pe");
    const int chunk = atoi(argv[6]);
    if (pe) opt.flag |= MEM_F_PE;

    bwams_index_t *idx = nullptr;
    int rc = bwams_index_open(prefix, 0, &idx);
    if (rc) { fprintf(stderr, "[bwams] %s: %s\n", bwams_strerror(rc), bwams_last_error()); return 1; }     // main_mem's convention
    {
        std::vector<bwams_contig_t> ctg;
        std::string names;
        std::vector<int32_t> noff{0};
        std::string spec = argv[7];
        size_t at = 0;
        while (at < spec.size()) {
            size_t e = spec.find(',', at);
            if (e == std::string::npos) e = spec.size();
            const std::string one = spec.substr(at, e - at);
            char nm[256];
            long long off, len; int alt;
            if (sscanf(one.c_str(), "%255[^:]:%lld:%lld:%d", nm, &off, &len, &alt) != 4) { fprintf(stderr, "bad contig spec\n"); return 2; }
            bwams_contig_t c; memset(&c, 0, sizeof c);
            c.offset = off; c.len = (int32_t)len; c.is_alt = alt;
            ctg.push_back(c);
            names += nm; names.push_back('\0'); noff.push_back((int32_t)names.size());
            at = e + 1;
        }
        if ((rc = bwams_index_set_contigs(idx, ctg.data(), (int32_t)ctg.size())) || (rc = bwams_index_set_contig_names(idx, names.data(), noff.data()))) {
            fprintf(stderr, "[bwams] %s: %s\n", bwams_strerror(rc), bwams_last_error()); return 1;
        }
    }
    // step 0: the chunk's records (four lines each) into bseq1_t, as bseq_read_orig leaves them (src/bwa.cpp:266-335)
    FILE *f = fopen(fq, "r");
    if (!f) { perror(fq); return 1; }
    std::vector<std::string> lines;
    {
        char *ln = nullptr; size_t cap = 0; ssize_t n;
        while ((n = getline(&ln, &cap, f)) > 0) { while (n > 0 && (ln[n - 1] == '\n' || ln[n - 1] == '\r')) ln[--n] = 0; lines.emplace_back(ln); }
        free(ln); fclose(f);
    }
    const int n_all = (int)(lines.size() / 4);
    bwams_worker *w = nullptr;
    if ((rc = bwams_worker_create(idx, nullptr, nullptr, chunk, (int64_t)chunk * 400, "", &w))) { fprintf(stderr, "[bwams] %s: %s\n", bwams_strerror(rc), bwams_last_error()); return 1; }
    FILE *fo = fopen(out, "w");
    int64_t n_processed = 0;
    for (int first = 0; first < n_all; first += chunk) {
        const int n = first + chunk < n_all ? chunk : n_all - first;
        std::vector<bseq1_t> seqs((size_t)n);
        std::vector<std::string> keep;
        keep.reserve((size_t)n * 4);
        for (int i = 0; i < n; ++i) {
            const std::string &h = lines[(size_t)(first + i) * 4];
            std::string name = h.substr(1), comment;
            const size_t sp = name.find_first_of(" \t");
            if (sp != std::string::npos) { comment = name.substr(sp + 1); name.resize(sp); }
            if (name.size() > 2 && name[name.size() - 2] == '/' && (name.back() == '1' || name.back() == '2')) name.resize(name.size() - 2);   // trim_readno
            keep.push_back(name); keep.push_back(lines[(size_t)(first + i) * 4 + 1]); keep.push_back(lines[(size_t)(first + i) * 4 + 3]);
            bseq1_t &s = seqs[(size_t)i];
            memset(&s, 0, sizeof s);
            s.name = &keep[keep.size() - 3][0]; s.seq = &keep[keep.size() - 2][0]; s.qual = &keep[keep.size() - 1][0];
            s.comment = nullptr;                    // process() frees the comments unless -C (src/fastmap.cpp:356-363)
            s.l_seq = (int)keep[keep.size() - 2].size();
            s.id = first + i;
        }
        mem_process_seqs(&opt, n_processed, n, seqs.data(), nullptr, *w);     // step 1
        for (int i = 0; i < n; ++i)                                            // step 2
            if (seqs[(size_t)i].sam) { fputs(seqs[(size_t)i].sam, fo); free(seqs[(size_t)i].sam); }
            else if (i % BATCH_SIZE == 0) { fprintf(stderr, "work item %d left no text\n", i); return 3; }
        n_processed += n;
    }
    fclose(fo);
    bwams_worker_destroy(w);
    bwams_index_close(idx);
    return 0;
}
