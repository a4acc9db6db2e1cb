// chunk_multi.cpp — one chunk over N batches (one per GPU, index replicated), behind ONE C call.
//
// The reference is one process calling mem_process_seqs once per chunk (src/fastmap.cpp:392-419), its work items fanned out over
// threads; mem_pestat runs over the WHOLE chunk between worker_aln and worker_sam (src/bwamem.cpp:1881-1891).  Here the chunk is cut
// into N contiguous shards on read (paired-end: pair) boundaries, one host thread drives each shard's batch:
//     upload (reads, names, qualities: PCIe only)  ->  stage 1 (worker_bwt + worker_aln) per shard  ->  [paired-end] the shards'
//     insert-size keys merged in-process, mem_pestat's loop over their union (bit-identical to the unsharded statistics: it depends on
//     the multiset of keys only)  ->  stage 2 (worker_sam) per shard with the chunk's statistics and the shard's first read / pair id
//     (the hash seeds of mem_mark_primary_se / mem_pair are global read ordinals, src/bwamem.cpp:1808-1810)  ->  the shards' SAM
//     texts back to back, i.e. in read order.
// No collective on the data path: the only exchange is the 8-byte keys, inside this process.
//
// Threads: a bwams_multi owns one worker thread per batch for its whole life (no thread is created per chunk); a call hands every
// thread one job and waits for all of them.  Nothing thrown inside a job or while the threads start crosses the C boundary.
// Kernels of two batches on ONE device do not interleave: every device has one compute lock in this process, taken around stages 1 and 2
// and not around uploads and fetches — so with two bwams_multi over the same devices chunk i + 1 goes up and chunk i - 1 comes down
// while chunk i computes (host/mem_process_seqs_hip.cpp), which is what the reference's `-i` pipeline threads overlap
// (src/fastmap.cpp:307-468, the ordering lock :475-491).
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bwams.h"

namespace {

// one compute lock per device, shared by every bwams_multi of the process
std::mutex &device_lock(int dev) {
    static std::mutex reg;
    static std::map<int, std::unique_ptr<std::mutex>> locks;
    std::lock_guard<std::mutex> g(reg);
    auto &p = locks[dev];
    if (!p) p.reset(new std::mutex());
    return *p;
}

struct Shard {
    bwams_batch_t *batch = nullptr;
    bwams_emf_t *emf = nullptr;
    bwams_ert_t *ert = nullptr;
    std::mutex *dev_lock = nullptr;
    // the shard's own offset arrays (start at 0), reused from chunk to chunk
    std::vector<int64_t> cum, noff, coff;
    std::vector<uint64_t> keys;
    int64_t bytes = 0;
    int rc = BWAMS_OK;
    std::string err;
    // the worker thread and its one-slot mailbox
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, quit = false, busy = false;
};

}  // namespace

struct bwams_multi {
    int n = 0;
    std::vector<std::unique_ptr<Shard>> sh;
    std::vector<int64_t> bounds;          // reads: shard s = [bounds[s], bounds[s + 1])
    int64_t n_reads = 0;
    int32_t paired = 0;
    bool uploaded = false, done = false;
    std::string err;

    // run f(s) on every shard's thread and wait for all of them
    void for_all(const std::function<void(int)> &f) {
        for (int s = 0; s < n; ++s) {
            Shard &x = *sh[(size_t)s];
            std::lock_guard<std::mutex> g(x.mu);
            x.job = [&f, s]() { f(s); };
            x.has_job = true;
            x.busy = true;
            x.cv.notify_all();
        }
        for (int s = 0; s < n; ++s) {
            Shard &x = *sh[(size_t)s];
            std::unique_lock<std::mutex> g(x.mu);
            x.cv.wait(g, [&x] { return !x.busy; });
        }
    }
    int first_error() {
        for (int s = 0; s < n; ++s)
            if (sh[(size_t)s]->rc) { err = "shard " + std::to_string(s) + ": " + sh[(size_t)s]->err; return sh[(size_t)s]->rc; }
        return BWAMS_OK;
    }
};

static void shard_main(Shard *x) {
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> g(x->mu);
            x->cv.wait(g, [x] { return x->has_job || x->quit; });
            if (x->quit && !x->has_job) return;
            job.swap(x->job);
            x->has_job = false;
        }
        try {
            job();
        } catch (const std::exception &e) {
            x->rc = BWAMS_ERR_NOMEM;
            x->err = e.what();
        } catch (...) {
            x->rc = BWAMS_ERR_NOMEM;
            x->err = "exception in a shard's job";
        }
        {
            std::lock_guard<std::mutex> g(x->mu);
            x->busy = false;
        }
        x->cv.notify_all();
    }
}

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

int bwams_multi_destroy(bwams_multi_t *m) {
    if (!m) return BWAMS_OK;
    for (auto &p : m->sh) {
        if (!p || !p->th.joinable()) continue;
        {
            std::lock_guard<std::mutex> g(p->mu);
            p->quit = true;
        }
        p->cv.notify_all();
        p->th.join();
    }
    delete m;
    return BWAMS_OK;
}

int bwams_multi_create(bwams_batch_t *const *batches, bwams_emf_t *const *emf, bwams_ert_t *const *ert, int32_t n, bwams_multi_t **out) {
    if (!batches || n < 1 || !out) return BWAMS_ERR_ARG;
    for (int i = 0; i < n; ++i)
        if (!batches[i]) return BWAMS_ERR_ARG;
    bwams_multi *m = nullptr;
    try {
        m = new bwams_multi();
        m->n = n;
        m->bounds.assign((size_t)n + 1, 0);
        for (int i = 0; i < n; ++i) {
            std::unique_ptr<Shard> x(new Shard());
            x->batch = batches[i];
            x->emf = emf ? emf[i] : nullptr;
            x->ert = ert ? ert[i] : nullptr;
            int32_t dev = 0;
            if (int rc = bwams_batch_device(batches[i], &dev)) { bwams_multi_destroy(m); return rc; }
            x->dev_lock = &device_lock(dev);
            m->sh.push_back(std::move(x));
        }
        for (int i = 0; i < n; ++i) m->sh[(size_t)i]->th = std::thread(shard_main, m->sh[(size_t)i].get());
    } catch (...) {                      // std::thread or an allocation: join what was started, report, throw nothing
        if (m) bwams_multi_destroy(m);
        return BWAMS_ERR_NOMEM;
    }
    *out = m;
    return BWAMS_OK;
}

const char *bwams_multi_error(const bwams_multi_t *m) { return m ? m->err.c_str() : ""; }

int bwams_multi_upload(bwams_multi_t *m, const uint8_t *enc_qdb, const int64_t *cum_len, int64_t n_reads, const char *names,
                       const int64_t *name_off, const char *quals, const char *comments, const int64_t *comment_off, int32_t paired) {
    if (!m || n_reads < 0 || (n_reads > 0 && (!enc_qdb || !cum_len || !names || !name_off)) || (comments && !comment_off)) return BWAMS_ERR_ARG;
    m->done = m->uploaded = false;
    m->n_reads = n_reads;
    m->paired = paired;
    m->err.clear();
    int rc = bwams_shard_bounds(n_reads, m->n, paired, m->bounds.data());
    if (rc) return rc;
    m->for_all([&](int s) {
        Shard &x = *m->sh[(size_t)s];
        x.rc = BWAMS_OK;
        x.bytes = 0;
        x.keys.clear();
        const int64_t lo = m->bounds[(size_t)s], k = m->bounds[(size_t)s + 1] - lo;
        x.cum.resize((size_t)k + 1);
        x.noff.resize((size_t)k + 1);
        x.coff.resize(comments ? (size_t)k + 1 : 0);
        for (int64_t i = 0; i <= k; ++i) {                      // the shard's offset arrays start at 0
            x.cum[(size_t)i] = cum_len[lo + i] - cum_len[lo];
            x.noff[(size_t)i] = name_off[lo + i] - name_off[lo];
            if (comments) x.coff[(size_t)i] = comment_off[lo + i] - comment_off[lo];
        }
        x.rc = bwams_process_reads_upload(x.batch, k ? enc_qdb + cum_len[lo] : nullptr, x.cum.data(), k, k ? names + name_off[lo] : nullptr,
                                          x.noff.data(), quals && k ? quals + cum_len[lo] : nullptr,
                                          comments && k ? comments + comment_off[lo] : nullptr, comments ? x.coff.data() : nullptr);
        if (x.rc) x.err = bwams_last_error();                   // the message lives in the thread that failed
    });
    if ((rc = m->first_error())) return rc;
    m->uploaded = true;
    return BWAMS_OK;
}

int bwams_multi_compute(bwams_multi_t *m, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo, const bwams_sam_opt_t *sam_opt,
                        const bwams_pestat_t *pes0, int64_t n_processed, int32_t flags, int64_t *sam_bytes) {
    if (!m || !so || !mo || !sam_opt || !m->uploaded) return BWAMS_ERR_ARG;
    const int32_t paired = m->paired;
    m->for_all([&](int s) {
        Shard &x = *m->sh[(size_t)s];
        const int64_t k = m->bounds[(size_t)s + 1] - m->bounds[(size_t)s];
        std::lock_guard<std::mutex> g(*x.dev_lock);
        int r = bwams_process_reads_stage1_run(x.batch, x.emf, x.ert, so, mo);
        if (!r && paired && !pes0 && k > 0) {
            x.keys.resize((size_t)(k / 2));
            int64_t nk = 0;
            r = bwams_pestat_keys(x.batch, mo, x.keys.data(), (int64_t)x.keys.size(), &nk);
            x.keys.resize((size_t)(r ? 0 : nk));
        }
        x.rc = r;
        if (r) x.err = bwams_last_error();
    });
    int rc = m->first_error();
    if (rc) return rc;
    bwams_pestat_t pes[4];
    memset(pes, 0, sizeof pes);
    if (paired) {
        if (pes0) memcpy(pes, pes0, sizeof pes);
        else {
            std::vector<uint64_t> all;
            for (int s = 0; s < m->n; ++s) all.insert(all.end(), m->sh[(size_t)s]->keys.begin(), m->sh[(size_t)s]->keys.end());
            if ((rc = bwams_pestat_from_keys(all.data(), (int64_t)all.size(), pes))) { m->err = bwams_last_error(); return rc; }
        }
    }
    m->for_all([&](int s) {
        Shard &x = *m->sh[(size_t)s];
        const int64_t lo = m->bounds[(size_t)s];
        const int64_t id_base = paired ? (n_processed >> 1) + (lo >> 1) : n_processed + lo;
        int64_t b = 0;
        std::lock_guard<std::mutex> g(*x.dev_lock);
        const int r = bwams_process_reads_stage2(x.batch, x.emf, x.ert, mo, sam_opt, paired, paired ? pes : nullptr, id_base, flags, &b);
        x.bytes = r ? 0 : b;
        x.rc = r;
        if (r) x.err = bwams_last_error();
    });
    if ((rc = m->first_error())) return rc;
    int64_t total = 0;
    for (int s = 0; s < m->n; ++s) total += m->sh[(size_t)s]->bytes;
    m->done = true;
    if (sam_bytes) *sam_bytes = total;
    return BWAMS_OK;
}

int bwams_multi_process_reads(bwams_multi_t *m, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo, const bwams_sam_opt_t *sam_opt,
                              const uint8_t *enc_qdb, const int64_t *cum_len, int64_t n_reads, const char *names, const int64_t *name_off,
                              const char *quals, const char *comments, const int64_t *comment_off, int32_t paired, const bwams_pestat_t *pes0,
                              int64_t n_processed, int32_t flags, int64_t *sam_bytes) {
    if (!m || !so || !mo || !sam_opt) return BWAMS_ERR_ARG;
    const int rc = bwams_multi_upload(m, enc_qdb, cum_len, n_reads, names, name_off, quals, comments, comment_off, paired);
    if (rc) return rc;
    return bwams_multi_compute(m, so, mo, sam_opt, pes0, n_processed, flags, sam_bytes);
}

// The chunk's SAM text in read order (the shards' texts back to back) and, per read, where its records start (n_reads + 1 offsets).
// Every shard's thread copies its own text (side by side over the devices' links; at the link's rate when `sam` is page-locked:
// bwams_host_alloc).
int bwams_multi_fetch(bwams_multi_t *m, char *sam, int64_t cap, int64_t *read_off) {
    if (!m || !m->done) return BWAMS_ERR_ARG;
    std::vector<int64_t> at((size_t)m->n + 1, 0);
    for (int s = 0; s < m->n; ++s) at[(size_t)s + 1] = at[(size_t)s] + m->sh[(size_t)s]->bytes;
    const int64_t total = at[(size_t)m->n];
    if (sam && total > cap) return BWAMS_ERR_CAPACITY;
    // shard s writes read_off[bounds[s] .. bounds[s + 1]] inclusive: the entry two shards share is written by both with the same
    // number only after the shift below, so every shard fetches into its own range first (the shared entry last: the shift is serial)
    m->for_all([&](int s) {
        Shard &x = *m->sh[(size_t)s];
        const int64_t lo = m->bounds[(size_t)s], k = m->bounds[(size_t)s + 1] - lo;
        x.rc = BWAMS_OK;
        if (k == 0) return;
        x.cum.resize((size_t)k + 1);                                     // reused as the shard's own offsets
        x.rc = bwams_sam_fetch(x.batch, sam ? sam + at[(size_t)s] : nullptr, x.bytes, read_off ? x.cum.data() : nullptr, nullptr, 0);
        if (x.rc) { x.err = bwams_last_error(); return; }
        if (read_off)
            for (int64_t i = 0; i < k; ++i) read_off[lo + i] = x.cum[(size_t)i] + at[(size_t)s];
    });
    const int rc = m->first_error();
    if (rc) return rc;
    if (read_off) read_off[m->n_reads] = total;
    return BWAMS_OK;
}
