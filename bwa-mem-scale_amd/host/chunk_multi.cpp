// chunk_multi.cpp — one chunk over N batches (one per GPU, index replicated), behind ONE C call.
//
// The reference is one process calling mem_process_seqs once per chunk (src/fastmap.cpp:392-419), its work items fanned out over
// threads; mem_pestat runs over the WHOLE chunk between worker_aln and worker_sam (src/bwamem.cpp:1881-1891).  Here the chunk is cut
// into N contiguous shards on read (paired-end: pair) boundaries, one host thread drives each shard's batch:
//     stage 1 (worker_bwt + worker_aln) per shard  ->  [paired-end] the shards' insert-size keys merged in-process, mem_pestat's
//     loop over their union (bit-identical to the unsharded statistics: it depends on the multiset of keys only)  ->  stage 2
//     (worker_sam) per shard with the chunk's statistics and the shard's first read / pair id (the hash seeds of
//     mem_mark_primary_se / mem_pair are global read ordinals, src/bwamem.cpp:1808-1810)  ->  the shards' SAM texts back to back,
//     i.e. in read order.
// No collective on the data path: the only exchange is the 8-byte keys, inside this process.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "bwams.h"

struct bwams_multi {
    int n = 0;
    std::vector<bwams_batch_t *> batch;
    std::vector<bwams_emf_t *> emf;
    std::vector<bwams_ert_t *> ert;
    std::vector<int64_t> bounds;          // reads: shard s = [bounds[s], bounds[s + 1])
    std::vector<int64_t> bytes;           // SAM bytes per shard of the last run
    int64_t n_reads = 0;
    bool done = false;
    std::string err;
};

// Shard s of a chunk of n_reads: sizes differ by at most one unit (a read, or a pair when paired), larger shards first
// (bwams/shard.py:shard_bounds is the same arithmetic for the Python ranks).
int bwams_shard_bounds(int64_t n_reads, int32_t n_shards, int32_t paired, int64_t *bounds) {
    if (n_reads < 0 || n_shards < 1 || !bounds || (paired && (n_reads & 1))) return BWAMS_ERR_ARG;
    const int64_t unit = paired ? 2 : 1, units = n_reads / unit;
    const int64_t base = units / n_shards, extra = units % n_shards;
    int64_t at = 0;
    for (int32_t s = 0; s < n_shards; ++s) {
        bounds[s] = at;
        at += (base + (s < extra ? 1 : 0)) * unit;
    }
    bounds[n_shards] = at;
    return BWAMS_OK;
}

int bwams_multi_create(bwams_batch_t *const *batches, bwams_emf_t *const *emf, bwams_ert_t *const *ert, int32_t n, bwams_multi_t **out) {
    if (!batches || n < 1 || !out) return BWAMS_ERR_ARG;
    bwams_multi *m = new bwams_multi();
    m->n = n;
    for (int i = 0; i < n; ++i) {
        if (!batches[i]) { delete m; return BWAMS_ERR_ARG; }
        m->batch.push_back(batches[i]);
        m->emf.push_back(emf ? emf[i] : nullptr);
        m->ert.push_back(ert ? ert[i] : nullptr);
    }
    m->bounds.assign((size_t)n + 1, 0);
    m->bytes.assign((size_t)n, 0);
    *out = m;
    return BWAMS_OK;
}

int bwams_multi_destroy(bwams_multi_t *m) { delete m; return BWAMS_OK; }

const char *bwams_multi_error(const bwams_multi_t *m) { return m ? m->err.c_str() : ""; }

int bwams_multi_process_reads(bwams_multi_t *m, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo, const bwams_sam_opt_t *sam_opt,
                              const uint8_t *enc_qdb, const int64_t *cum_len, int64_t n_reads, const char *names, const int64_t *name_off,
                              const char *quals, const char *comments, const int64_t *comment_off, int32_t paired, const bwams_pestat_t *pes0,
                              int64_t n_processed, int32_t flags, int64_t *sam_bytes) {
    if (!m || !so || !mo || !sam_opt || n_reads < 0 || (n_reads > 0 && (!enc_qdb || !cum_len || !names || !name_off)) || (comments && !comment_off))
        return BWAMS_ERR_ARG;
    const int N = m->n;
    m->done = false;
    m->n_reads = n_reads;
    m->err.clear();
    int rc = bwams_shard_bounds(n_reads, N, paired, m->bounds.data());
    if (rc) return rc;
    std::vector<int> rcs((size_t)N, BWAMS_OK);
    std::vector<std::string> errs((size_t)N);
    std::vector<std::vector<uint64_t>> keys((size_t)N);

    auto stage1 = [&](int s) {
        const int64_t lo = m->bounds[s], hi = m->bounds[s + 1], k = hi - lo;
        // the shard's offset arrays start at 0
        std::vector<int64_t> cum((size_t)k + 1), noff((size_t)k + 1), coff(comments ? (size_t)k + 1 : 0);
        for (int64_t i = 0; i <= k; ++i) {
            cum[(size_t)i] = cum_len[lo + i] - cum_len[lo];
            noff[(size_t)i] = name_off[lo + i] - name_off[lo];
            if (comments) coff[(size_t)i] = comment_off[lo + i] - comment_off[lo];
        }
        int r = bwams_process_reads_stage1(m->batch[s], m->emf[s], m->ert[s], so, mo, enc_qdb + cum_len[lo], cum.data(), k,
                                           names + name_off[lo], noff.data(), quals ? quals + cum_len[lo] : nullptr,
                                           comments ? comments + comment_off[lo] : nullptr, comments ? coff.data() : nullptr);
        if (!r && paired && !pes0 && k > 0) {
            keys[s].resize((size_t)(k / 2));
            int64_t nk = 0;
            r = bwams_pestat_keys(m->batch[s], mo, keys[s].data(), (int64_t)keys[s].size(), &nk);
            keys[s].resize((size_t)(r ? 0 : nk));
        }
        rcs[s] = r;
        if (r) errs[s] = bwams_last_error();      // the message lives in the thread that failed
    };
    {
        std::vector<std::thread> th;
        for (int s = 1; s < N; ++s) th.emplace_back(stage1, s);
        stage1(0);
        for (auto &t : th) t.join();
    }
    for (int s = 0; s < N; ++s)
        if (rcs[s]) { m->err = "shard " + std::to_string(s) + ": " + errs[s]; return rcs[s]; }

    bwams_pestat_t pes[4];
    memset(pes, 0, sizeof pes);
    if (paired) {
        if (pes0) memcpy(pes, pes0, sizeof pes);
        else {
            std::vector<uint64_t> all;
            for (int s = 0; s < N; ++s) all.insert(all.end(), keys[s].begin(), keys[s].end());
            if ((rc = bwams_pestat_from_keys(all.data(), (int64_t)all.size(), pes))) { m->err = bwams_last_error(); return rc; }
        }
    }
    auto stage2 = [&](int s) {
        const int64_t lo = m->bounds[s];
        const int64_t id_base = paired ? (n_processed >> 1) + (lo >> 1) : n_processed + lo;
        int64_t b = 0;
        const int r = bwams_process_reads_stage2(m->batch[s], m->emf[s], m->ert[s], mo, sam_opt, paired, paired ? pes : nullptr, id_base, flags, &b);
        m->bytes[s] = r ? 0 : b;
        rcs[s] = r;
        if (r) errs[s] = bwams_last_error();
    };
    {
        std::vector<std::thread> th;
        for (int s = 1; s < N; ++s) th.emplace_back(stage2, s);
        stage2(0);
        for (auto &t : th) t.join();
    }
    int64_t total = 0;
    for (int s = 0; s < N; ++s) {
        if (rcs[s]) { m->err = "shard " + std::to_string(s) + ": " + errs[s]; return rcs[s]; }
        total += m->bytes[s];
    }
    m->done = true;
    if (sam_bytes) *sam_bytes = total;
    return BWAMS_OK;
}

// The chunk's SAM text in read order (the shards' texts back to back) and, per read, where its records start (n_reads + 1 offsets).
int bwams_multi_fetch(bwams_multi_t *m, char *sam, int64_t cap, int64_t *read_off) {
    if (!m || !m->done) return BWAMS_ERR_ARG;
    int64_t total = 0;
    for (int s = 0; s < m->n; ++s) total += m->bytes[s];
    if (sam && total > cap) return BWAMS_ERR_CAPACITY;
    std::vector<int> rcs((size_t)m->n, BWAMS_OK);
    std::vector<int64_t> at((size_t)m->n + 1, 0);
    for (int s = 0; s < m->n; ++s) at[(size_t)s + 1] = at[(size_t)s] + m->bytes[s];
    auto one = [&](int s) {
        const int64_t lo = m->bounds[s], k = m->bounds[s + 1] - lo;
        if (k == 0) return;
        rcs[s] = bwams_sam_fetch(m->batch[s], sam ? sam + at[(size_t)s] : nullptr, m->bytes[s], read_off ? read_off + lo : nullptr, nullptr, 0);
        if (!rcs[s] && read_off && at[(size_t)s])
            for (int64_t i = 0; i <= k; ++i) read_off[lo + i] += at[(size_t)s];       // the shard's offsets start at 0
    };
    // in shard order: shard s writes read_off[bounds[s] .. bounds[s + 1]] inclusive, and the next shard overwrites the shared end
    // with its own start — the same number
    for (int s = 0; s < m->n; ++s) one(s);
    for (int s = 0; s < m->n; ++s) if (rcs[s]) return rcs[s];
    if (read_off) read_off[m->n_reads] = total;
    return BWAMS_OK;
}
